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
//   P4  match resolution by all waves, 4 KiB of the image at a time: per byte the distance to an
//       ancestor, pointer jumping until every byte points at a final one, one parallel copy
//       (src/inflate.ts:287-290)
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

#define RES_W 4096u    // P4: bytes of the image resolved at a time (a multiple of PAR_THREADS and of 8)
#define P4_SPARSE_DEP_MAX 32u  // ... of which at most this many copy from another match's bytes
#define P4_SPARSE_MAX 1024u  // P4 (T1): a block with at most this many matches is copied match by match
#ifndef RES_REPS
#define RES_REPS 4u    // pointer-jumping steps between two barriers
#endif
#define PAR_CHUNK (ZES_BLK - 3u * RES_W)  // T2: bytes of a block's output resolved per pass over the image; behind them: room for a match record at the chunk's last byte and P4's distance array (the decode tables stay alive for the next chunk)
#define PAR_DIST_OFF (PAR_CHUNK + 64u)
#define PAR_MAX_OUT (1u << 27)      // T2: longest block taken (output bytes)
#define STAGE_DW ((ZES_BLK + ZES_BLK / 8) / 4)  // compressed bytes staged over the image + bitmap until P3
struct ParSmem {
  uint8_t out[ZES_BLK];            // P0-P2: first part of the staged compressed block (swizzled dwords)
  uint32_t bitmap[ZES_BLK / 32];   // P0-P2: rest of the staging area; cleared before P3
  uint32_t lut_l[1u << PL_ROOT];
  uint32_t lut_d[1u << PD_ROOT];
  uint8_t len_l[1u << PL_ROOT];  // bits of the lit/len part of a token: [5:0] count, 0x40 distance follows, 0x80 use lut_l
  uint8_t len_d[1u << PD_ROOT];  // bits of the distance part: [5:0] count, 0x80 use lut_d
  uint16_t syms_l[288];
  uint16_t syms_d[32];
  uint32_t first_l[16], first_d[16];
  uint16_t cnt_l[16], cnt_d[16], offs_l[16], offs_d[16];
  uint8_t lens[352];
  uint8_t cl_lut[128];
  uint32_t wave_sum[PAR_WAVES];
  uint8_t wtab[PAR_WAVES][48];   // composed transfer table of each wave
  uint8_t gtab[PAR_WAVES][4][48];  // ... of each wave's four groups of sixteen segments
  uint8_t wentry[PAR_WAVES];      // entry code of each wave's first segment
  uint32_t hdr_end, status, tail_entry, bfinal, tail_bytes, tail_end;
  uint32_t h_hlit, h_total, h_wbase, h_off, h_k, h_prev, h_done;  // the header's second part: what its rounds hand on
  uint32_t res_flag[3];       // P4: "a pointer moved in this step"
  uint32_t res_lastw[RES_W / 32];  // P4: last match start (+1) at or before each 32 positions of the window
  uint32_t res_strad[4];      // P4: end and distance of the match that runs into the window from the one before; the next one's
  uint32_t f8lo, f8n, f8off;  // 8-bit literal codes: first code value, how many (0 = fast path off), index into syms_l
  uint32_t sp_n;  // P4, few matches: how many
};

static_assert(offsetof(ParSmem, cl_lut) - offsetof(ParSmem, lut_l) >= RES_W * 2u, "P4's distance array lies over the decode tables");
static_assert(RES_W % PAR_THREADS == 0 && RES_W % 8u == 0, "P4 lane mapping");

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

// Where a lane's bits come from: the swizzled LDS copy of the block (P0-P2) or global memory.
struct BitSrc {
  const uint32_t* g32;  // the buffer as dwords
  uint32_t lastdw;      // last valid dword of the buffer
  const uint32_t* s32;  // LDS staging area
  uint32_t s_first;     // global dword index held at staging slot 0
  uint32_t s_count;     // staged dwords
};
template <bool LDS>
__device__ __forceinline__ static uint32_t src_ldw(const BitSrc& s, uint32_t idx) {
  if (LDS) {
    const uint32_t k = idx - s.s_first;
    // lanes read at a stride of one segment (~32 dwords for random data): XOR the row index in
    return k < s.s_count ? s.s32[k ^ ((k >> 5) & 31u)] : 0u;
  }
  return s.g32[idx < s.lastdw ? idx : s.lastdw];
}

struct LaneBits {
  uint64_t bb;
  uint32_t nb;
  uint32_t pos;  // bit offset of bb's bit 0 inside the buffer; (pos + nb) % 32 == 0
  uint32_t pre;  // reading global memory: the dword behind bb's bits, dword (pos + nb) / 32, asked for one refill early
};
template <bool LDS>
__device__ __forceinline__ static void lb_seek(LaneBits& b, const BitSrc& s, uint32_t bit) {
  const uint32_t i = bit >> 5, sh = bit & 31u;
  const uint64_t w = (uint64_t)src_ldw<LDS>(s, i) | ((uint64_t)src_ldw<LDS>(s, i + 1) << 32);
  b.bb = w >> sh;
  b.nb = 64u - sh;
  b.pos = bit;
  b.pre = LDS ? 0u : src_ldw<LDS>(s, i + 2);
}
// From global memory (the emit pass: the LDS holds the output image by then) a refill takes the dword that was asked for
// at the refill before and asks for the next one: the wave does not sit out a trip to memory every 32 bits of its
// slowest lane (the emit pass was 130k cycles per block on random data for ~40k cycles of instructions).
template <bool LDS>
__device__ __forceinline__ static void lb_refill(LaneBits& b, const BitSrc& s) {
  if (b.nb <= 32u) {
    if (LDS) {
      b.bb |= (uint64_t)src_ldw<LDS>(s, (b.pos + b.nb) >> 5) << b.nb;
      b.nb += 32u;
    } else {
      b.bb |= (uint64_t)b.pre << b.nb;
      b.nb += 32u;
      b.pre = src_ldw<LDS>(s, (b.pos + b.nb) >> 5);
    }
  }
}
__device__ __forceinline__ static uint32_t lb_take(LaneBits& b, uint32_t k) {
  const uint32_t v = (uint32_t)b.bb & ((1u << k) - 1u);
  b.bb >>= k;
  b.nb -= k;
  b.pos += k;
  return v;
}

// Incompressible data gives nearly every literal an 8-bit code.  Canonical codes of one length
// are a contiguous range of code values and the literals come first in it, so "the next four
// tokens are 8-bit literals" is four range tests on the bit-reversed dword: no table lookups.
struct Lit8 {
  uint32_t lo, n, off;  // n == 0: fast path off for this block
};
// how many of the next four tokens (0..4, counted from the first) are 8-bit literals
__device__ __forceinline__ static uint32_t lead_lit8(uint32_t w, const Lit8& f) {
  const uint32_t r = __brev(w);
  const bool a = ((r >> 24) - f.lo) < f.n, b = (((r >> 16) & 255u) - f.lo) < f.n, c = (((r >> 8) & 255u) - f.lo) < f.n,
             d = ((r & 255u) - f.lo) < f.n;
  uint32_t n = d ? 4u : 3u;
  n = c ? n : 2u;
  n = b ? n : 1u;
  n = a ? n : 0u;
  return n;
}

// transfer table of a segment in registers: 64 entries x 8 bits.  Eight named words and explicit
// select chains on purpose: an indexed array makes the compiler move the table to scratch.
struct SegTab {
  uint64_t a, b, c, d, e, f, g, h;
};
__device__ __forceinline__ static uint32_t tab_get(const SegTab& t, uint32_t o) {
  const uint32_t wi = o >> 3;
  // opaque copies: otherwise the select chain over fields is folded back into an indexed load
  uint64_t a = t.a, b = t.b, c = t.c, d = t.d, e = t.e, f = t.f, g = t.g, h = t.h;
  asm("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  asm("" : "+v"(e), "+v"(f), "+v"(g), "+v"(h));
  uint64_t x = a;
  x = (wi == 1u) ? b : x;
  x = (wi == 2u) ? c : x;
  x = (wi == 3u) ? d : x;
  x = (wi == 4u) ? e : x;
  x = (wi == 5u) ? f : x;
  x = (wi == 6u) ? g : x;
  x = (wi == 7u) ? h : x;
  return (uint32_t)(x >> ((o & 7u) * 8u)) & 0xffu;
}
__device__ __forceinline__ static void tab_set(SegTab& t, uint32_t o, uint32_t code) {  // entries start as 0, set once
  const uint32_t wi = o >> 3;
  const uint64_t v = (uint64_t)code << ((o & 7u) * 8u);
  t.a |= (wi == 0u) ? v : 0ull;
  t.b |= (wi == 1u) ? v : 0ull;
  t.c |= (wi == 2u) ? v : 0ull;
  t.d |= (wi == 3u) ? v : 0ull;
  t.e |= (wi == 4u) ? v : 0ull;
  t.f |= (wi == 5u) ? v : 0ull;
  t.g |= (wi == 6u) ? v : 0ull;
  t.h |= (wi == 7u) ? v : 0ull;
}
__device__ __forceinline__ static void tab_put(SegTab& t, uint32_t o, uint32_t code, bool doit) {  // replaces an entry
  const uint32_t wi = o >> 3, sh = (o & 7u) * 8u;
  const uint64_t m = doit ? (0xffull << sh) : 0ull, v = doit ? ((uint64_t)code << sh) : 0ull;
  t.a = (wi == 0u) ? ((t.a & ~m) | v) : t.a;
  t.b = (wi == 1u) ? ((t.b & ~m) | v) : t.b;
  t.c = (wi == 2u) ? ((t.c & ~m) | v) : t.c;
  t.d = (wi == 3u) ? ((t.d & ~m) | v) : t.d;
  t.e = (wi == 4u) ? ((t.e & ~m) | v) : t.e;
  t.f = (wi == 5u) ? ((t.f & ~m) | v) : t.f;
  t.g = (wi == 6u) ? ((t.g & ~m) | v) : t.g;
  t.h = (wi == 7u) ? ((t.h & ~m) | v) : t.h;
}
// bit o: entry o is not 0
__device__ __forceinline__ static uint64_t tab_nzmask(const SegTab& t) {
  auto nz8 = [](uint64_t w) -> uint64_t {
    const uint64_t k7 = 0x7f7f7f7f7f7f7f7full;
    const uint64_t y = (((w & k7) + k7) | w) & ~k7;             // bit 7 of every byte that is not 0
    return ((y >> 7) * 0x0102040810204080ull) >> 56;            // ... gathered into one byte
  };
  return nz8(t.a) | (nz8(t.b) << 8) | (nz8(t.c) << 16) | (nz8(t.d) << 24) | (nz8(t.e) << 32) | (nz8(t.f) << 40) | (nz8(t.g) << 48) | (nz8(t.h) << 56);
}
__device__ __forceinline__ static uint64_t bcast64(uint64_t v, uint32_t srclane) {  // srclane uniform
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)srclane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)srclane);
  return (uint64_t)lo | ((uint64_t)hi << 32);
}
__device__ __forceinline__ static SegTab tab_bcast(const SegTab& t, uint32_t srclane) {
  SegTab r;
  r.a = bcast64(t.a, srclane);
  r.b = bcast64(t.b, srclane);
  r.c = bcast64(t.c, srclane);
  r.d = bcast64(t.d, srclane);
  r.e = bcast64(t.e, srclane);
  r.f = bcast64(t.f, srclane);
  r.g = bcast64(t.g, srclane);
  r.h = bcast64(t.h, srclane);
  return r;
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
template <bool LDS>
__device__ __forceinline__ static uint32_t tok_step(ParSmem& S, LaneBits& b, const BitSrc& src, uint32_t& val, uint32_t& len,
                                                    uint32_t& dist) {
  lb_refill<LDS>(b, src);
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
  lb_refill<LDS>(b, src);
  len = (e >> 16) + lb_take(b, (e >> 4) & 15u);
  lb_refill<LDS>(b, src);
  uint32_t ed = S.lut_d[(uint32_t)b.bb & ((1u << PD_ROOT) - 1u)];
  if ((ed & 15u) == 0u) {
    uint32_t l2 = 0;
    const int sy = slow_sym(b, PD_ROOT, S.syms_d, S.first_d, S.cnt_d, S.offs_d, &l2);
    if (sy < 0) return T_FAIL;
    ed = entry_d((uint32_t)sy, l2);
  }
  lb_take(b, ed & 15u);
  if (((ed >> 8) & 3u) != 2u) return T_FAIL;
  lb_refill<LDS>(b, src);
  dist = (ed >> 16) + lb_take(b, (ed >> 4) & 15u);
  return T_MATCH;
}

// code of one litlen LUT hit (slow path for codes longer than the root); 0 = no code matches
__device__ __forceinline__ static uint32_t lut_l_entry(ParSmem& S, uint32_t bits) {
  uint32_t e = S.lut_l[bits & ((1u << PL_ROOT) - 1u)];
  if ((e & 15u) == 0u) {
    LaneBits t;
    t.bb = bits;
    uint32_t l2 = 0;
    const int sy = slow_sym(t, PL_ROOT, S.syms_l, S.first_l, S.cnt_l, S.offs_l, &l2);
    e = sy < 0 ? 0u : entry_l((uint32_t)sy, l2);
  }
  return e;
}
__device__ __forceinline__ static uint32_t lut_d_entry(ParSmem& S, uint32_t bits) {
  uint32_t e = S.lut_d[bits & ((1u << PD_ROOT) - 1u)];
  if ((e & 15u) == 0u) {
    LaneBits t;
    t.bb = bits;
    uint32_t l2 = 0;
    const int sy = slow_sym(t, PD_ROOT, S.syms_d, S.first_d, S.cnt_d, S.offs_d, &l2);
    e = sy < 0 ? 0u : entry_d((uint32_t)sy, l2);
  }
  return e;
}

// branch-free refill from the staged copy (a step costs the same whether or not a lane needs bits)
__device__ __forceinline__ static void lb_refill_bf(LaneBits& b, const BitSrc& s) {
  const bool need = b.nb <= 32u;
  const uint32_t k = ((b.pos + b.nb) >> 5) - s.s_first;
  const bool in = k < s.s_count;
  const uint32_t kk = in ? k : 0u;
  uint32_t w = s.s32[kk ^ ((kk >> 5) & 31u)];
  w = in ? w : 0u;
  const uint32_t sh = need ? b.nb : 0u;
  const uint64_t add = (uint64_t)w << sh;
  b.bb |= need ? add : 0ull;
  b.nb += need ? 32u : 0u;
}

// Bit length of the token at b.pos for every lane (positions only, no values): byte LUTs first,
// the full tables only behind a ballot (EOB, codes longer than the root, bad symbols).
// On return b is positioned so that consuming `adv` more bits completes the token.
__device__ __forceinline__ static void len_step(ParSmem& S, const BitSrc& src, LaneBits& b, bool act, uint32_t& adv, bool& eob,
                                                bool& bad) {
  lb_refill_bf(b, src);
  const uint32_t a = S.len_l[(uint32_t)b.bb & ((1u << PL_ROOT) - 1u)];
  uint32_t n = a & 63u;
  const bool sp = (a & 0x80u) != 0u;
  bool m = (a & 0x40u) != 0u;
  eob = false;
  bad = false;
  if (__ballot(act && sp)) {
    const uint32_t e = lut_l_entry(S, (uint32_t)b.bb);
    const uint32_t kind = (e >> 8) & 3u;
    if (sp) {
      eob = kind == 1u;
      bad = (kind == 3u) || ((e & 15u) == 0u);
      n = (e & 15u) + ((e >> 4) & 15u);
      m = kind == 2u;
    }
  }
  if (__ballot(act && m)) {
    LaneBits t = b;
    t.bb >>= n;
    t.nb -= n;
    t.pos += n;
    lb_refill_bf(t, src);
    const uint32_t d = S.len_d[(uint32_t)t.bb & ((1u << PD_ROOT) - 1u)];
    uint32_t n2 = d & 63u;
    const bool dsp = (d & 0x80u) != 0u;
    if (__ballot(act && m && dsp)) {
      const uint32_t ed = lut_d_entry(S, (uint32_t)t.bb);
      if (dsp) {
        bad = bad || (m && ((((ed >> 8) & 3u) != 2u) || (ed & 15u) == 0u));
        n2 = (ed & 15u) + ((ed >> 4) & 15u);
      }
    }
    if (act && m) {
      b = t;
      n = n2;
    }
  }
  adv = n;
}

// ------------------------------------------------------------------------------------------
// Transfer table of the segment [base, stop), branch-light:
//  (a) for each of the 64 window offsets o: where does ONE token starting at base+o end?
//      (64 independent decodes from a register copy of the first 160 bits, fully unrolled)
//  (b) tokens that end past the window land on at most 48 distinct offsets; only from those a
//      full trajectory to the segment end is decoded (highest first, so a trajectory that
//      reaches a landing point already done takes over its result)
//  (c) table[o] = table[successor] filled from offset 63 down, every write at a static slot.
// ------------------------------------------------------------------------------------------
#define NX_EOB 0xFEu
#define NX_FAIL 0xFFu

// 32 bits at relative bit r (dynamic, r < 128) of the aligned window a0..a4
__device__ __forceinline__ static uint32_t win_bits(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t r) {
  const uint32_t wi = r >> 5;
  uint32_t lo = a0, hi = a1;
  lo = (wi == 1u) ? a1 : lo;
  hi = (wi == 1u) ? a2 : hi;
  lo = (wi == 2u) ? a2 : lo;
  hi = (wi == 2u) ? a3 : hi;
  lo = (wi == 3u) ? a3 : lo;
  hi = (wi == 3u) ? a4 : hi;
  return __builtin_amdgcn_alignbit(hi, lo, r & 31u);
}

// ------------------------------------------------------------------------------------------
// Transfer table of a segment when nearly every literal has an 8-bit code (incompressible data).
// There the eight bit phases never merge, and the generic construction below decodes the segment
// eight times, token by token.  Instead:
//  (A) one regular pass over the segment's dwords tests every bit position for "an 8-bit literal
//      code starts here" — four positions of one phase per SWAR range test on the bit-reversed
//      dword.  A token chain on phase f (positions = f mod 8) runs through such positions in steps
//      of eight; only where the test fails (other code lengths, matches, end of block: about nine
//      positions per segment on random data, the *listed* positions, kept in a 32-entry list in
//      registers) it can do anything else;
//  (B) ONE fold over the listed positions, highest first, each ONE token decode.  E[f] = exit code
//      of a chain that is on phase f at the fold's position (eight bytes in a register pair; at the
//      segment's end: the first position of the phase at or behind it).  A token from q to q'
//      gives q the code q' - end, or end-of-block / fail, or E[q' mod 8] — every listed position
//      above q is folded into E already — and E[q mod 8] becomes that code.  (Round 2 walked a
//      trajectory per phase and entry from list search to list search: every listed position
//      decoded by each chain that meets it, the wave at the pace of its longest chain per
//      trajectory, 36 list slots compared per search.)
//      The one thing E cannot tell: listed positions p between q and q' on the phase of q' are
//      folded in although the token jumps over them (half the listed positions are matches of 20-40
//      bits: it happens several times per block).  The positions folded last are at hand — the
//      list is taken four at a time — so they are counted: none -> E; one -> E1, the value E[f]
//      had before its latest change (that change was p); two -> E2; three, or more positions
//      under one token than the look-back holds -> the lane gives up and its wave redoes the
//      generic construction.
//      The table: entry o = E[o mod 8] as of position o, i.e. word w of the table (entries
//      8w..8w+7) is E once every listed position >= 8w is folded in.
// Returns false for a lane that must use the generic construction (list overflow, data end near).
// ------------------------------------------------------------------------------------------
#define F8_MAXSEG 1800u  // longest segment (bits) the 11-bit list entries and the scan loop are sized for
#define F8_LIST 32u

__device__ __forceinline__ static bool seg_table_f8(ParSmem& S, const BitSrc& src, uint32_t limit, uint32_t base, uint32_t stop,
                                                    const Lit8& f8, SegTab& tab, unsigned long long* dp) {
  // debug stamps (ZES_DEBUG_PHASES): slots 8..10 by the first wave of the workgroup, 12..14 by the last one
#define T8STAMP(i)                                                                  \
  do {                                                                              \
    if (dp && (threadIdx.x == 0 || threadIdx.x == PAR_THREADS - 64))                \
      dp[(threadIdx.x ? 12 : 8) + (i)] = (unsigned long long)clock64();             \
  } while (0)
  const uint32_t sl = stop - base;
  const uint32_t sl_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)sl);
  // (every bit the scan looks at is data; the segments of a block have one length)
  bool ok = sl == sl_u && sl >= 64u && sl <= F8_MAXSEG && stop + 128u <= limit;
  uint32_t why = ok ? 0u : 1u;  // ZES_DEBUG_PHASES: what made the lane give up
  // ---- (A) positions where no 8-bit literal code starts ----
  // per byte of the bit-reversed dword: t = byte - lo (mod 256); the byte is a literal code iff t < n.
  // With m = 256 - n <= 128: t >= n  <=>  bit7(t) and bit7((t & 0x7f) + m)   (no carries between bytes)
  const uint32_t H = 0x80808080u;
  const uint32_t LO = f8.lo * 0x01010101u, M = (256u - f8.n) * 0x01010101u;
  const uint32_t LOm = LO & ~H, nLO = ~LO;
  // incompressible data proper: the 8-bit codes start at code value 0 and all but a few (< 8) of the 256 byte
  // patterns are literals: "no literal here" is a run of five one bits and a three-bit comparison — bitwise on the
  // dword, 32 positions at once (20 operations per dword; the byte-wise range test above: 93)
  const bool runs = f8.lo == 0u && f8.n >= 248u;  // (uniform; n <= 255: the end-of-block code needs room too)
  const uint32_t nc0 = (f8.n & 1u) ? 0u : ~0u, nc1 = (f8.n & 2u) ? 0u : ~0u, nc2 = (f8.n & 4u) ? 0u : ~0u;
  // thirty-two u16 positions relative to the segment base, four per word, 0xFFFF = empty; the newest (highest) four in
  // grp, newest in its low 16 bits.  Named words and no index on purpose (see SegTab).
  uint64_t q0 = ~0ull, q1 = ~0ull, q2 = ~0ull, q3 = ~0ull, q4 = ~0ull, q5 = ~0ull, q6 = ~0ull, q7 = ~0ull, grp = ~0ull;
  uint32_t nl = 0;
  const uint32_t d0 = base >> 5, bo = base & 31u;
  const uint32_t nd = (min(sl_u, F8_MAXSEG) + 62u) >> 5;  // dwords that cover [base, stop) for any alignment (uniform)
  const uint32_t ebit = bo + sl;                           // first bit behind the segment, counted from dword d0
  uint32_t lo = src_ldw<true>(src, d0);
  // four dwords per step: the listing loop below runs as often as the busiest lane has positions to list,
  // so it is entered once per 128 bit positions, not once per 32
#pragma unroll 1
  for (uint32_t j0 = 0; j0 < nd; j0 += 4u) {
    uint32_t bad[4];
    if (runs) {  // (uniform)
#pragma unroll
      for (uint32_t jj = 0; jj < 4u; jj++) {
        const uint32_t hi = src_ldw<true>(src, d0 + j0 + jj + 1u);
        // the bit-reversed byte at p is >= n = 0b11111ccc: stream bits p..p+4 are ones and bits p+5..p+7, read as a
        // number most significant first, are >= ccc (compared from the last bit up: with the constant's bit set the
        // stream's bit must be set and the rest hold, with it clear either will do)
        const uint32_t s1 = __builtin_amdgcn_alignbit(hi, lo, 1), s2 = __builtin_amdgcn_alignbit(hi, lo, 2),
                       s3 = __builtin_amdgcn_alignbit(hi, lo, 3), s4 = __builtin_amdgcn_alignbit(hi, lo, 4),
                       s5 = __builtin_amdgcn_alignbit(hi, lo, 5), s6 = __builtin_amdgcn_alignbit(hi, lo, 6),
                       s7 = __builtin_amdgcn_alignbit(hi, lo, 7);
        uint32_t ge = s7 | nc0;
        ge = ((s6 ^ ge) & nc1) | (s6 & ge);
        ge = ((s5 ^ ge) & nc2) | (s5 & ge);
        bad[jj] = lo & s1 & s2 & s3 & s4 & ge;
        lo = hi;
      }
    } else {
#pragma unroll
      for (uint32_t jj = 0; jj < 4u; jj++) {
        const uint32_t hi = src_ldw<true>(src, d0 + j0 + jj + 1u);
        uint32_t bd = 0;  // bit b: no 8-bit literal code starts at bit 32 j + b
#pragma unroll
        for (uint32_t phi = 0; phi < 8u; phi++) {
          const uint32_t x = __brev(__builtin_amdgcn_alignbit(hi, lo, phi));  // top byte = the token at bit 32 j + phi
          const uint32_t t = ((x | H) - LOm) ^ ((x ^ nLO) & H);
          const uint32_t inv = t & ((t & ~H) + M) & H;  // bit 31 - 8k: token k of this phase
          bd |= (__brev(inv) << phi);                    // -> bit phi + 8k
        }
        bad[jj] = bd;
        lo = hi;
      }
    }
#pragma unroll
    for (uint32_t jj = 0; jj < 4u; jj++) {
      // only positions of the segment: bits [bo, bo + sl) counted from dword d0
      const uint32_t j = j0 + jj;
      const int32_t eb = (int32_t)ebit - (int32_t)(32u * j);
      uint32_t vm = eb >= 32 ? ~0u : eb <= 0 ? 0u : ((1u << (eb & 31)) - 1u);
      vm &= (j == 0u) ? (~0u << bo) : ~0u;
      bad[jj] &= vm;
    }
    while (__ballot((bad[0] | bad[1] | bad[2] | bad[3]) != 0u)) {
      // lowest position first: the list ascends, its newest entry is the highest position
      const uint32_t q = bad[0] ? 0u : bad[1] ? 1u : bad[2] ? 2u : 3u;
      const uint32_t bq = bad[0] ? bad[0] : bad[1] ? bad[1] : bad[2] ? bad[2] : bad[3];
      const bool take = bq != 0u;
      const uint32_t bpos = take ? (uint32_t)__builtin_ctz(bq) : 0u;
      const uint32_t cl = bq & (bq - 1u);  // that bit cleared
      bad[0] = (q == 0u) ? cl : bad[0];
      bad[1] = (q == 1u && take) ? cl : bad[1];
      bad[2] = (q == 2u && take) ? cl : bad[2];
      bad[3] = (q == 3u && take) ? cl : bad[3];
      const uint32_t rel = 32u * (j0 + q) + bpos - bo;
      // a complete group of four goes to the list when the next position arrives (grp is only empty for an empty list):
      // the words move up by one; past 32 positions the oldest drop out, and the lane falls back anyway.  No branch and
      // no indexed word: both cost this loop two dozen register copies per turn.
      const bool full = take && nl != 0u && (nl & 3u) == 0u;
      q7 = full ? q6 : q7;
      q6 = full ? q5 : q6;
      q5 = full ? q4 : q5;
      q4 = full ? q3 : q4;
      q3 = full ? q2 : q3;
      q2 = full ? q1 : q2;
      q1 = full ? q0 : q1;
      q0 = full ? grp : q0;
      grp = full ? ~0ull : grp;
      grp = take ? ((grp << 16) | rel) : grp;
      nl += take ? 1u : 0u;
    }
  }
  ok = ok && nl <= F8_LIST + 4u;
  why |= nl <= F8_LIST + 4u ? 0u : 2u;
  T8STAMP(0);
  // ---- (B) the fold ----
  uint64_t E = 0;
#pragma unroll
  for (uint32_t f = 0; f < 8u; f++) E |= (uint64_t)((f - sl) & 7u) << (8u * f);
  uint64_t E1 = E, E2 = E;
  uint64_t T1 = E, T2 = E, T3 = E, T4 = E, T5 = E;  // table words 1..5: E when the fold has passed entry 8w
  uint64_t cw = grp;  // the four positions in hand
#pragma unroll 1
  while (__ballot(cw != ~0ull)) {
#pragma unroll
    for (uint32_t sidx = 0; sidx < 4u; sidx++) {
      const uint32_t q = (uint32_t)(cw >> (16u * sidx)) & 0xFFFFu;
      const bool act = q != 0xFFFFu;
      // the token at q: 64 bits from there on, bit lengths from the byte tables (the full tables behind a ballot)
      const uint32_t ab = base + (act ? q : 0u);
      const uint32_t di = ab >> 5, sh = ab & 31u;
      const uint32_t w0 = src_ldw<true>(src, di), w1 = src_ldw<true>(src, di + 1u), w2 = src_ldw<true>(src, di + 2u);
      const uint32_t b0 = __builtin_amdgcn_alignbit(w1, w0, sh), b1 = __builtin_amdgcn_alignbit(w2, w1, sh);
      const uint32_t a = S.len_l[b0 & ((1u << PL_ROOT) - 1u)];
      uint32_t n = a & 63u;
      bool m = (a & 0x40u) != 0u, eob = false, bd2 = false;
      const bool sp = (a & 0x80u) != 0u;
      if (__ballot(act && sp)) {
        const uint32_t e = lut_l_entry(S, b0);
        const uint32_t kind = (e >> 8) & 3u;
        if (sp) {
          eob = kind == 1u;
          bd2 = (kind == 3u) || ((e & 15u) == 0u);
          n = (e & 15u) + ((e >> 4) & 15u);
          m = kind == 2u;
        }
      }
      if (__ballot(act && m)) {
        const uint32_t tb = (uint32_t)(((((uint64_t)b1) << 32) | b0) >> (n & 31u));  // n <= 20
        const uint32_t dd = S.len_d[tb & ((1u << PD_ROOT) - 1u)];
        uint32_t n2 = dd & 63u;
        const bool dsp = (dd & 0x80u) != 0u;
        if (__ballot(act && m && dsp)) {
          const uint32_t ed = lut_d_entry(S, tb);
          if (dsp) {
            bd2 = bd2 || (m && ((((ed >> 8) & 3u) != 2u) || (ed & 15u) == 0u));
            n2 = (ed & 15u) + ((ed >> 4) & 15u);
          }
        }
        n += m ? n2 : 0u;
      }
      const uint32_t x = q + n;  // where the token ends (1..48 bits on)
      // listed positions it jumps over on the phase it lands on: x - 8, x - 16, .. above q.  Whether one is listed is
      // the test of (A) on the 64 bits in hand: the bytes of the window from bit (n mod 8) on are the candidates
      uint32_t nov = 0;
      if (__ballot(act && n >= 9u)) {
        const uint64_t y = ((((uint64_t)b1) << 32) | b0) >> (n & 7u);
        const uint32_t r0 = __brev((uint32_t)y), r1 = __brev((uint32_t)(y >> 32));  // byte k of the window: top byte k of r0, r1
        const uint32_t t0 = ((r0 | H) - LOm) ^ ((r0 ^ nLO) & H), t1 = ((r1 | H) - LOm) ^ ((r1 ^ nLO) & H);
        const uint32_t i0 = t0 & ((t0 & ~H) + M) & H, i1 = t1 & ((t1 & ~H) + M) & H;
        // candidates: bytes kmin..kmax, kmin = 1 when the window starts at q itself, kmax = n / 8 - 1
        const uint32_t kmin = (n & 7u) ? 0u : 1u, kmax = (n >> 3) - 1u;  // (n >= 9 here: kmax >= 0; n <= 48: kmax <= 5)
        const uint64_t im = (((uint64_t)i0) << 32) | i1;                 // byte k flagged at bit 63 - 8k
        const uint64_t vm = (~0ull >> (8u * kmin)) & (~0ull << (56u - 8u * (kmax & 7u)));
        nov = (n >= 9u) ? (uint32_t)__popcll(im & vm) : 0u;
      }
      const uint32_t fx = (x & 7u) * 8u;
      uint64_t ev = nov == 0u ? E : E1;
      ev = nov == 2u ? E2 : ev;
      uint32_t code = x >= sl ? x - sl : ((uint32_t)(ev >> fx) & 0xffu);
      code = bd2 ? C_FAIL : code;
      code = eob ? C_EOB : code;
      ok = ok && !(act && !eob && !bd2 && x < sl && nov >= 3u);
      why |= (act && !eob && !bd2 && x < sl && nov >= 3u) ? 4u : 0u;
      const uint32_t fq = (q & 7u) * 8u;
      const uint64_t fm = act ? (0xffull << fq) : 0ull;
      E2 = (E2 & ~fm) | (E1 & fm);
      E1 = (E1 & ~fm) | (E & fm);
      E = (E & ~fm) | (((uint64_t)code << fq) & fm);
      T1 = (act && q >= 8u) ? E : T1;
      T2 = (act && q >= 16u) ? E : T2;
      T3 = (act && q >= 24u) ? E : T3;
      T4 = (act && q >= 32u) ? E : T4;
      T5 = (act && q >= 40u) ? E : T5;
    }
    cw = q0;
    q0 = q1;
    q1 = q2;
    q2 = q3;
    q3 = q4;
    q4 = q5;
    q5 = q6;
    q6 = q7;
    q7 = ~0ull;
  }
  T8STAMP(1);
  tab.a = E;
  tab.b = T1;
  tab.c = T2;
  tab.d = T3;
  tab.e = T4;
  tab.f = T5;
  tab.g = 0;
  tab.h = 0;
  T8STAMP(2);
  if (dp) {
    const uint64_t w1 = __ballot((why & 3u) != 0u), w4 = __ballot((why & 4u) != 0u), w8 = 0;
    if (zes_lane() == 0) {
      if (w1) atomicAdd(&dp[29], (unsigned long long)__popcll(w1));
      if (w4) atomicAdd(&dp[30], (unsigned long long)__popcll(w4));
      if (w8) atomicAdd(&dp[31], (unsigned long long)__popcll(w8));
    }
  }
#undef T8STAMP
  return ok;
}

// ------------------------------------------------------------------------------------------
// Transfer table of a segment of compressible data, when there is room behind the staged block (round 3): the
// exit code of EVERY bit position, last position first.  code[p] = where the token that starts at p ends, if that is
// behind the segment (or end-of-block / fail), else code[p + its length] — a position the sweep has been at.  One
// token-LENGTH decode per position (two byte lookups: 13-bit and 12-bit tables built for this sweep, so that codes
// longer than the root tables' 10 / 9 bits, which garbage positions hit all the time, stay off the slow path), one
// byte read and one byte write of the lane's 64-byte ring in LDS; the ring's first 48 bytes are the segment's table
// when the sweep arrives at position 0.  ~35 instructions per bit position, every lane the same number of steps —
// the window-and-trajectory construction below decodes fewer positions (~100 + 7 chains x 25 tokens per segment of
// text) but at the pace of the slowest lane of each step: ~120 instructions per bit position on the text workload.
// ------------------------------------------------------------------------------------------
#define DP_ROW 68u          // bytes per lane: 64 ring entries, 17 dwords apart (odd: the lanes' slots fall into different banks)
#define DP_LROOT 13u
#define DP_DROOT 12u
#define DP_BYTES (PAR_THREADS * DP_ROW + (1u << DP_LROOT) + (1u << DP_DROOT))

// the two byte tables (all threads): what len_l / len_d hold, for DP_LROOT / DP_DROOT bits of lookahead
__device__ __forceinline__ static void dp_build(ParSmem& S, uint8_t* T13, uint8_t* D12) {
  for (uint32_t i = threadIdx.x; i < (1u << DP_LROOT); i += PAR_THREADS) {
    uint32_t a = S.len_l[i & ((1u << PL_ROOT) - 1u)];
    if (a == 0x80u) {
      const uint32_t e = lut_l_entry(S, i);
      const uint32_t len = e & 15u, kind = (e >> 8) & 3u, bl = len + ((e >> 4) & 15u);
      a = (len == 0u || len > DP_LROOT) ? 0x80u : (kind == 0u) ? bl : (kind == 2u) ? (bl | 0x40u) : 0x80u;
    }
    T13[i] = (uint8_t)a;
  }
  for (uint32_t i = threadIdx.x; i < (1u << DP_DROOT); i += PAR_THREADS) {
    uint32_t a = S.len_d[i & ((1u << PD_ROOT) - 1u)];
    if (a == 0x80u) {
      const uint32_t e = lut_d_entry(S, i);
      const uint32_t len = e & 15u, kind = (e >> 8) & 3u, bl = len + ((e >> 4) & 15u);
      a = (len == 0u || len > DP_DROOT || kind != 2u) ? 0x80u : bl;
    }
    D12[i] = (uint8_t)a;
  }
}

// (segments of the sweep start on dword boundaries and are a multiple of 32 bits long: every lane is at the same bit of
// its dword and the same ring slot, so positions need no per-lane bookkeeping)
__device__ __forceinline__ static void seg_table_dp(ParSmem& S, const BitSrc& src, uint32_t limit, uint32_t base, uint32_t seglen,
                                                    const uint8_t* T13, const uint8_t* D12, uint8_t* row) {
  const uint32_t d0 = base >> 5;
  const uint32_t nd = seglen >> 5;
  const uint32_t room = limit > base ? limit - base : 0u;  // bits of data from base on
  // (uniform) some lane's segment ends within 64 bits of the data's end, or behind it: tokens are checked against it
  const bool near = __ballot((uint64_t)base + seglen + 64u > (uint64_t)limit) != 0ull;
  uint32_t mid = src_ldw<true>(src, d0 + nd), hi = src_ldw<true>(src, d0 + nd + 1u);
  // Three positions in flight inside a dword (the sweep goes from bit 31 down to bit 0): A(q) asks for position q's lit/len
  // entry, B(q) takes it, finds the distance part's place and asks for that entry, C(q) finishes the token and moves the
  // ring.  An iteration runs A(bb-2), B(bb-1), C(bb): both table reads of a position are on their way one and two turns
  // before they are used, and only the ring's read-then-write is waited for.  (Before: three dependent LDS round trips per
  // position, ~300 of its ~700 cycles.)  The pipeline drains at the dword's end: two positions in 32 without the overlap.
  uint32_t w_b = 0, a_b = 0;                              // A -> B: the window, the lit/len entry
  uint32_t n_c = 0, ovr_c = 0, dd_c = 0, t_c = 0;         // B -> C
  bool m_c = false;
  auto stage_a = [&](uint32_t lo, uint32_t q) {
    w_b = __builtin_amdgcn_alignbit(mid, lo, q);
    a_b = T13[w_b & ((1u << DP_LROOT) - 1u)];
  };
  auto stage_b = [&](uint32_t q) {
    const uint32_t w32 = w_b, a = a_b;
    const uint32_t wh = __builtin_amdgcn_alignbit(hi, mid, q);
    uint32_t n = a & 63u, ovr = 0;
    bool m = (a & 0x40u) != 0u;
    const bool sp = (a & 0x80u) != 0u;
    if (__ballot(sp)) {  // end of block, a code longer than 13 bits, no code
      const uint32_t e = lut_l_entry(S, w32);
      const uint32_t kind = (e >> 8) & 3u;
      if (sp) {
        ovr = (kind == 1u) ? C_EOB : ((kind == 3u) || ((e & 15u) == 0u)) ? C_FAIL : 0u;
        n = (e & 15u) + ((e >> 4) & 15u);
        m = kind == 2u;
      }
    }
    const uint32_t t = __builtin_amdgcn_alignbit(wh, w32, n);  // n <= 20
    n_c = n;
    ovr_c = ovr;
    m_c = m;
    t_c = t;
    dd_c = D12[t & ((1u << DP_DROOT) - 1u)];
  };
#pragma unroll 1
  for (uint32_t j = nd; j-- > 0u;) {
    const uint32_t lo = src_ldw<true>(src, d0 + j);
    stage_a(lo, 31u);
    stage_b(31u);
    stage_a(lo, 30u);
#ifndef DP_UNROLL
#define DP_UNROLL 8
#endif
#pragma unroll DP_UNROLL
    for (uint32_t bb = 32u; bb-- > 0u;) {
      const uint32_t p = 32u * j + bb;
      // C(bb): what B(bb) left
      uint32_t n = n_c, ovr = ovr_c;
      const bool m = m_c;
      const uint32_t dd = dd_c, t = t_c;
      // B(bb-1) (its entry was asked for a turn ago), then A(bb-2)
      if (bb >= 1u) stage_b(bb - 1u);
      if (bb >= 2u) stage_a(lo, bb - 2u);
      uint32_t n2 = dd & 63u;
      const bool dsp = m && (dd & 0x80u) != 0u;
      if (__ballot(dsp)) {
        const uint32_t ed = lut_d_entry(S, t);
        if (dsp) {
          ovr = ((((ed >> 8) & 3u) != 2u) || (ed & 15u) == 0u) ? C_FAIL : ovr;
          n2 = (ed & 15u) + ((ed >> 4) & 15u);
        }
      }
      n += m ? n2 : 0u;
      const uint32_t x = p + n;  // 1..48 bits on
      const uint32_t r = row[x & 63u];
      uint32_t code = x < seglen ? r : x - seglen;
      code = ovr ? ovr : code;
      if (near) code = x > room ? C_FAIL : code;  // the token runs past the data
      row[p & 63u] = (uint8_t)code;
    }
    hi = mid;
    mid = lo;
  }
}

template <bool LDS, bool TWO>
__device__ __forceinline__ static void seg_table(ParSmem& S, const BitSrc& src, uint32_t limit, uint32_t base, uint32_t stop,
                                                 const Lit8& f8, SegTab& tab, unsigned long long* dp) {
  // uniform: most literals have 8-bit codes, and the segments are long enough for literal runs to matter
  // (a block of long matches has 64-bit segments that are nearly all listed positions)
  if (LDS && f8.n && !__ballot(stop - base < 256u)) {
    const bool done = seg_table_f8(S, src, limit, base, stop, f8, tab, dp);
    const uint64_t redo = __ballot(!done);
    if (dp && redo && zes_lane() == 0) {  // ZES_DEBUG_PHASES: lanes / waves that fell back
      atomicAdd(&dp[11], (unsigned long long)__popcll(redo));
      atomicAdd(&dp[15], 1ull);
    }
    if (!redo) return;  // a wave with a lane that could not use it redoes the segment tables the generic way
  }
  // debug stamps (ZES_DEBUG_PHASES): slots 8..10 by the first wave of the workgroup, 12..14 by the last one
#define TSTAMP(i)                                                                   \
  do {                                                                              \
    if (dp && (threadIdx.x == 0 || threadIdx.x == PAR_THREADS - 64))                \
      dp[(threadIdx.x ? 12 : 8) + (i)] = (unsigned long long)clock64();             \
  } while (0)
  // ---- (a) successors of the 64 window offsets ----
  uint32_t a0, a1, a2, a3, a4;
  {
    const uint32_t d0 = base >> 5, bo = base & 31u;
    const uint32_t w0 = src_ldw<LDS>(src, d0), w1 = src_ldw<LDS>(src, d0 + 1), w2 = src_ldw<LDS>(src, d0 + 2),
                   w3 = src_ldw<LDS>(src, d0 + 3), w4 = src_ldw<LDS>(src, d0 + 4), w5 = src_ldw<LDS>(src, d0 + 5);
    a0 = __builtin_amdgcn_alignbit(w1, w0, bo);
    a1 = __builtin_amdgcn_alignbit(w2, w1, bo);
    a2 = __builtin_amdgcn_alignbit(w3, w2, bo);
    a3 = __builtin_amdgcn_alignbit(w4, w3, bo);
    a4 = __builtin_amdgcn_alignbit(w5, w4, bo);
  }
  SegTab nx = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t Lmask = 0;  // landing offsets 64..111 (bit = offset - 64)
  const uint32_t room = limit > base ? limit - base : 0u;  // bits of data from base on
#pragma unroll 1
  for (uint32_t o = 0; o < 64u; o++) {  // a loop, not unrolled: the body must stay small in the instruction cache
    const uint32_t e = lut_l_entry(S, win_bits(a0, a1, a2, a3, a4, o));
    const uint32_t kind = (e >> 8) & 3u;
    uint32_t v = o + (e & 15u) + ((e >> 4) & 15u);
    if (__ballot(kind == 2u)) {
      const uint32_t ed = lut_d_entry(S, win_bits(a0, a1, a2, a3, a4, v & 127u));
      const uint32_t v2 = v + (ed & 15u) + ((ed >> 4) & 15u);
      const bool dbad = (((ed >> 8) & 3u) != 2u) || (ed & 15u) == 0u;
      v = (kind == 2u) ? (dbad ? NX_FAIL : v2) : v;
    }
    v = (kind == 1u) ? NX_EOB : v;
    v = (kind == 3u || (e & 15u) == 0u) ? NX_FAIL : v;
    v = (v < 128u && v > room) ? NX_FAIL : v;
    v = (o >= room) ? NX_FAIL : v;
    Lmask |= (v >= 64u && v < 128u) ? (1ull << (v - 64u)) : 0ull;
    tab_set(nx, o, v);  // v >= 1: a token has at least one bit
  }

  TSTAMP(0);
  // ---- (a2) long segments: a second window.  Of its 64 offsets only those are decoded that a token of the first
  //      window lands on, or a token of the second window that starts on such an offset, and so on: about 30.  The
  //      trajectories of (b) then start 128 bits into the segment, where the 48 entry offsets have merged into ~7
  //      token chains instead of ~12 — and every one of them costs the wave the longest lane's walk to the end. ----
  const bool two = TWO && LDS && !__ballot(stop - base < 176u);  // (uniform: the segments of a block have one length)
  SegTab nx2 = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t Lmask2 = 0;
  if (two) {
    uint32_t b0, b1, b2, b3, b4;
    {
      const uint32_t d0 = (base + 64u) >> 5, bo = (base + 64u) & 31u;
      const uint32_t w0 = src_ldw<LDS>(src, d0), w1 = src_ldw<LDS>(src, d0 + 1), w2 = src_ldw<LDS>(src, d0 + 2),
                     w3 = src_ldw<LDS>(src, d0 + 3), w4 = src_ldw<LDS>(src, d0 + 4), w5 = src_ldw<LDS>(src, d0 + 5);
      b0 = __builtin_amdgcn_alignbit(w1, w0, bo);
      b1 = __builtin_amdgcn_alignbit(w2, w1, bo);
      b2 = __builtin_amdgcn_alignbit(w3, w2, bo);
      b3 = __builtin_amdgcn_alignbit(w4, w3, bo);
      b4 = __builtin_amdgcn_alignbit(w5, w4, bo);
    }
    const uint32_t room2 = room > 64u ? room - 64u : 0u;
    uint64_t todo = Lmask;  // offsets of the second window still to decode (bit r: position base + 64 + r), lowest first
#pragma unroll 1
    while (__ballot(todo != 0ull)) {
      const bool hv = todo != 0ull;
      const uint32_t r = hv ? (uint32_t)__builtin_ctzll(todo) : 0u;
      todo &= todo - 1ull;  // (0 stays 0)
      const uint32_t e = lut_l_entry(S, win_bits(b0, b1, b2, b3, b4, r));
      const uint32_t kind = (e >> 8) & 3u;
      uint32_t v = r + (e & 15u) + ((e >> 4) & 15u);
      if (__ballot(hv && kind == 2u)) {
        const uint32_t ed = lut_d_entry(S, win_bits(b0, b1, b2, b3, b4, v & 127u));
        const uint32_t v2 = v + (ed & 15u) + ((ed >> 4) & 15u);
        const bool dbad = (((ed >> 8) & 3u) != 2u) || (ed & 15u) == 0u;
        v = (kind == 2u) ? (dbad ? NX_FAIL : v2) : v;
      }
      v = (kind == 1u) ? NX_EOB : v;
      v = (kind == 3u || (e & 15u) == 0u) ? NX_FAIL : v;
      v = (v < 128u && v > room2) ? NX_FAIL : v;
      v = (r >= room2) ? NX_FAIL : v;
      if (hv) {
        todo |= (v < 64u) ? (1ull << v) : 0ull;  // (v > r: still ahead)
        Lmask2 |= (v >= 64u && v < 128u) ? (1ull << (v - 64u)) : 0ull;
        tab_set(nx2, r, v);
      }
    }
  }
  const uint32_t ZO = two ? 128u : 64u;  // where the landing window of the trajectories starts
  const uint64_t Lm = two ? Lmask2 : Lmask;
  // ---- (b) full trajectories from the landing offsets, highest first ----
  SegTab lt = {0, 0, 0, 0, 0, 0, 0, 0};  // exit code of landing offset ZO+i at entry i
  uint64_t Ldone = 0;
  for (;;) {
    const uint64_t pend = Lm & ~Ldone;
    const bool have = pend != 0ull;
    if (!__ballot(have)) break;
    const uint32_t L = have ? 63u - (uint32_t)__builtin_clzll(pend) : 0u;
    LaneBits b;
    lb_seek<LDS>(b, src, base + ZO + L);
    uint32_t code = C_FAIL;
    bool act = have && (base + ZO + L) < limit;
    // (the landing-window check only matters while a trajectory is still inside that window)
    while (__ballot(act)) {
      // up to twelve 8-bit literals at once, past the landing window and inside the segment (skipped
      // as a whole while every lane is still in its landing window: most trajectories end there)
      if (f8.n && __ballot(act && (b.pos - base) >= ZO + 48u)) {
#pragma unroll
        for (int rep = 0; rep < 3; rep++) {
          lb_refill_bf(b, src);
          const bool fast = act && (b.pos - base) >= ZO + 48u && b.pos + 32u <= stop && b.pos + 32u <= limit;
          const uint32_t adv = fast ? 8u * lead_lit8((uint32_t)b.bb, f8) : 0u;
          b.bb >>= adv;
          b.nb -= adv;
          b.pos += adv;
        }
      }
      if (act && b.pos >= stop) {
        code = b.pos - stop;  // 0..47
        act = false;
      }
      const uint32_t rel = b.pos - base - ZO;  // landing-window offset
      const bool inwin = act && rel < 48u;
      if (__ballot(inwin)) {
        const bool hit = inwin && ((Ldone >> rel) & 1ull);
        if (__ballot(hit)) {
          if (hit) {
            code = tab_get(lt, rel);
            act = false;
          }
        }
      }
      uint32_t adv;
      bool eob, bad;
      len_step(S, src, b, act, adv, eob, bad);
      adv = act ? adv : 0u;
      b.bb >>= adv;
      b.nb -= adv;
      b.pos += adv;
      if (act && eob) {
        code = C_EOB;
        act = false;
      }
      if (act && (bad || b.pos > limit)) {
        code = C_FAIL;
        act = false;
      }
    }
    if (have) {
      tab_set(lt, L, code);
      Ldone |= 1ull << L;
    }
  }

  TSTAMP(1);
  // ---- (c2) the decoded offsets of the second window, highest first, in place: successor -> exit code ----
  if (two) {
    uint64_t m = tab_nzmask(nx2);  // the decoded offsets (a successor is never 0)
#pragma unroll 1
    while (__ballot(m != 0ull)) {
      const bool hv = m != 0ull;
      const uint32_t r = hv ? 63u - (uint32_t)__builtin_clzll(m) : 0u;
      m &= ~(hv ? (1ull << r) : 0ull);
      const uint32_t v = tab_get(nx2, r);
      const uint32_t in = tab_get(nx2, v & 63u);  // (v > r: converted already)
      const uint32_t la = tab_get(lt, (v - 64u) & 63u);
      uint32_t c = (v < 64u) ? in : la;
      c = (v == NX_EOB) ? C_EOB : c;
      c = (v == NX_FAIL) ? C_FAIL : c;
      tab_put(nx2, r, c, hv);
    }
    lt = nx2;  // what a token of the first window lands on
  }
  // ---- (c) table[o] from offset 63 down: successor inside the window -> its entry; landing ->
  //          the landing's code ----
  SegTab t = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
  for (int o = 63; o >= 0; o--) {
    const uint32_t v = tab_get(nx, (uint32_t)o);
    const uint32_t in = tab_get(t, v & 63u);
    const uint32_t la = tab_get(lt, (v - 64u) & 63u);
    uint32_t c = (v < 64u) ? in : la;
    c = (v == NX_EOB) ? C_EOB : c;
    c = (v == NX_FAIL) ? C_FAIL : c;
    tab_set(t, (uint32_t)o, c);
  }
  tab = t;
  TSTAMP(2);
#undef TSTAMP
}

// Decodes tokens from bit `entry` while the token start is below `stop`.
template <bool EMIT, bool LDS, bool HIST = false>
__device__ __forceinline__ static void seg_decode(ParSmem& S, const BitSrc& src, uint32_t limit, uint32_t entry, uint32_t stop,
                                                  uint32_t out_off, uint32_t& exit_pos, uint32_t& outbytes, uint32_t& flags,
                                                  uint32_t clo = 0) {
  // (HIST, the any-encoder form: output positions are clipped to the chunk [clo, clo + PAR_CHUNK) of the image)
  LaneBits b;
  lb_seek<LDS>(b, src, entry);
  uint32_t ob = 0, fl = 0;
  while (b.pos < stop) {
    uint32_t v = 0, len = 0, dist = 0;
    const uint32_t kind = tok_step<LDS>(S, b, src, v, len, dist);
    if (kind == T_FAIL || b.pos > limit) {  // (an end-of-block code whose bits lie behind the data counts as well:
      fl |= F_FAIL;                         //  the reference throws 'Lack of data length' there)
      break;
    }
    if (kind == T_EOB) {
      fl |= F_EOB;
      break;
    }
    if (kind == T_LIT) {
      if (EMIT) {
        const uint32_t q = out_off + ob - clo;
        if (!HIST || q < PAR_CHUNK) S.out[q] = (uint8_t)v;
      }
      ob++;
    } else {
      if (EMIT) {
        const uint32_t p = out_off + ob, q = p - clo;
        if (!HIST && dist > p) {
          fl |= F_HIST;  // looks behind the block start: not a reference-made block
        } else if (!HIST || q < PAR_CHUNK) {
          S.out[q] = (uint8_t)(dist - 1u);
          S.out[q + 1] = (uint8_t)((dist - 1u) >> 8);
          S.out[q + 2] = (uint8_t)(len - 3u);
          atomicOr(&S.bitmap[q >> 5], 1u << (q & 31u));
        } else if (p < clo && p + len > clo) {  // the match that runs into the chunk from the one before
          S.res_strad[0] = p + len - clo;
          S.res_strad[1] = dist;
        }
      }
      ob += len;
    }
    if (ob > (HIST ? PAR_MAX_OUT : ZES_BLK)) {  // more than a slot: not a reference-made block
      fl |= F_FAIL;
      break;
    }
  }
  exit_pos = b.pos;
  outbytes = ob;
  flags = fl;
}

// Predicated form of seg_decode for the count (P2) and emit (P3) passes: one loop for the whole
// wave, lanes drop out by clearing `act`; the rarely needed distance lookup sits behind a ballot.
template <bool EMIT, bool LDS, bool HIST = false>
__device__ __forceinline__ static void seg_run(ParSmem& S, const BitSrc& src, uint32_t limit, uint32_t entry, uint32_t stop, bool live,
                                               const Lit8& f8, uint32_t out_off, uint32_t& exit_pos, uint32_t& outbytes,
                                               uint32_t& flags, uint32_t clo = 0) {
  LaneBits b;
  lb_seek<LDS>(b, src, live ? entry : 0u);
  uint32_t ob = 0, fl = 0;
  bool act = live;
  // T1's emit pass: literals leave as whole dwords.  The lanes' output ranges lie ~128 bytes apart — one LDS bank for
  // half the wave — so every store instruction of the wave costs ~32 turns of that bank, and a byte store per literal
  // made the stores the pass's bottleneck (126k cycles per block of random data).  A lane collects its bytes and stores
  // the aligned dwords that belong to it alone; bytes in front of its first dword boundary, behind its last one, and
  // around a match record go one by one.
  // (only where the fast path for 8-bit literals is on — incompressible data, the regular 128-byte stride: on text the
  // lanes' ranges are irregular, the byte stores spread over the banks, and collecting cost more than it saved)
  const bool ACC = EMIT && !HIST && f8.n != 0u;  // (uniform)
  uint32_t pend = 0, pn = 0;  // bytes not stored yet (positions out_off + ob - pn .. out_off + ob - 1, the first one on a dword boundary)
  auto put = [&](uint32_t v, uint32_t n) __attribute__((always_inline)) {  // n <= 4 bytes of v, lowest first, behind the ones so far
    uint32_t P = out_off + ob;  // where the first of them goes
    v &= n >= 4u ? ~0u : ((1u << (8u * n)) - 1u);
    if (pn == 0u && (P & 3u) != 0u && n != 0u) {  // up to the lane's first dword boundary: byte by byte
      const uint32_t k = min(n, 4u - (P & 3u));
      for (uint32_t i = 0; i < k; i++) S.out[P + i] = (uint8_t)(v >> (8u * i));
      v = k < 4u ? v >> (8u * k) : 0u;
      n -= k;
      P += k;
    }
    const uint64_t w = (uint64_t)pend | ((uint64_t)v << (8u * pn));
    const uint32_t t = pn + n;
    if (t >= 4u) *reinterpret_cast<uint32_t*>(S.out + (P - pn)) = (uint32_t)w;
    pend = t >= 4u ? (uint32_t)(w >> 32) : (uint32_t)w;
    pn = t >= 4u ? t - 4u : t;
  };
  auto flush = [&]() __attribute__((always_inline)) {  // the bytes in hand, one by one (a match record or the lane's end follows)
    const uint32_t P = out_off + ob - pn;
    for (uint32_t i = 0; i < pn; i++) S.out[P + i] = (uint8_t)(pend >> (8u * i));
    pend = 0;
    pn = 0;
  };
  while (__ballot(act)) {
    if (f8.n) {  // uniform: four 8-bit literals at a time while they stay inside the segment
#pragma unroll
      for (int rep = 0; rep < 3; rep++) {
        lb_refill<LDS>(b, src);
        const uint32_t w = (uint32_t)b.bb;
        const bool fast = act && b.pos + 32u <= stop && b.pos + 32u <= limit && ob + 4u <= (HIST ? PAR_MAX_OUT : ZES_BLK);
        const uint32_t nlit = fast ? lead_lit8(w, f8) : 0u;
        if (EMIT && nlit) {
          // symbols of all four candidates are looked up (clamped into the table), the stores are per byte:
          // a byte past the literals belongs to a later token, possibly of the neighbouring lane
          const uint32_t r = __brev(w);
          const uint32_t k0 = f8.off - f8.lo;  // wraps when off < lo; the sums below are in range again
          const uint32_t i0 = k0 + (r >> 24), i1 = k0 + ((r >> 16) & 255u), i2 = k0 + ((r >> 8) & 255u), i3 = k0 + (r & 255u);
          const uint32_t s0 = S.syms_l[i0], s1 = S.syms_l[nlit > 1u ? i1 : i0], s2 = S.syms_l[nlit > 2u ? i2 : i0],
                         s3 = S.syms_l[nlit > 3u ? i3 : i0];
          if (ACC) {
            put((s0 & 255u) | ((s1 & 255u) << 8) | ((s2 & 255u) << 16) | (s3 << 24), nlit);
          } else {
            uint8_t* o = S.out + out_off + ob;
            o[0] = (uint8_t)s0;
            if (nlit > 1u) o[1] = (uint8_t)s1;
            if (nlit > 2u) o[2] = (uint8_t)s2;
            if (nlit > 3u) o[3] = (uint8_t)s3;
          }
        }
        const uint32_t adv = 8u * nlit;
        ob += nlit;
        b.bb >>= adv;
        b.nb -= adv;
        b.pos += adv;
      }
    }
    if (act && b.pos >= stop) act = false;
    lb_refill<LDS>(b, src);
    const uint32_t e = lut_l_entry(S, (uint32_t)b.bb);
    const uint32_t kind = (e >> 8) & 3u;
    const uint32_t cl = e & 15u, xb = (e >> 4) & 15u;
    uint32_t adv = cl + xb;
    bool bad = (kind == 3u) || (cl == 0u);
    uint32_t len = (e >> 16) + (((uint32_t)(b.bb >> cl)) & ((1u << xb) - 1u));  // length value of a match token
    uint32_t dist = 0;
    const bool ismatch = act && kind == 2u;
    if (__ballot(ismatch)) {
      LaneBits t = b;
      t.bb >>= adv;
      t.nb -= adv;
      t.pos += adv;
      lb_refill<LDS>(t, src);
      const uint32_t ed = lut_d_entry(S, (uint32_t)t.bb);
      const uint32_t dl = ed & 15u, dx = (ed >> 4) & 15u;
      dist = (ed >> 16) + (((uint32_t)(t.bb >> dl)) & ((1u << dx) - 1u));
      bad = bad || (kind == 2u && ((((ed >> 8) & 3u) != 2u) || dl == 0u));
      if (ismatch) {  // never move a lane that has already stopped: its b.pos is the segment's exit
        b = t;
        adv = dl + dx;
      }
    }
    adv = act ? adv : 0u;
    b.bb >>= adv;
    b.nb -= adv;
    b.pos += adv;
    if (act && (bad || b.pos > limit)) {  // (before the end-of-block test: its code must lie inside the data too)
      fl |= F_FAIL;
      act = false;
    }
    if (act && kind == 1u) {
      fl |= F_EOB;
      act = false;
    }
    if (act) {
      if (kind == 0u) {
        if (ACC) {
          put((e >> 16) & 255u, 1u);
        } else if (EMIT) {
          const uint32_t q = out_off + ob - clo;
          if (!HIST || q < PAR_CHUNK) S.out[q] = (uint8_t)(e >> 16);
        }
        ob += 1u;
      } else {
        if (ACC) flush();
        if (EMIT) {
          const uint32_t p = out_off + ob, q = p - clo;
          if (!HIST && dist > p) {
            fl |= F_HIST;  // looks behind the block start: not a reference-made block
          } else if (!HIST || q < PAR_CHUNK) {
            S.out[q] = (uint8_t)(dist - 1u);
            S.out[q + 1] = (uint8_t)((dist - 1u) >> 8);
            S.out[q + 2] = (uint8_t)(len - 3u);
            atomicOr(&S.bitmap[q >> 5], 1u << (q & 31u));
          } else if (p < clo && p + len > clo) {  // the match that runs into the chunk from the one before
            S.res_strad[0] = p + len - clo;
            S.res_strad[1] = dist;
          }
        }
        ob += len;
      }
      if (ob > (HIST ? PAR_MAX_OUT : ZES_BLK)) {  // more than a slot: not a reference-made block
        fl |= F_FAIL;
        act = false;
      }
    }
  }
  if (ACC) flush();
  exit_pos = b.pos;
  outbytes = ob;
  flags = fl;
}

// Canonical arrays of one alphabet (lens at S.lens[base..base+nsym)): the symbols in (length, symbol) order, first code,
// count and offset per length.  One WAVE per chunk of 64 symbols (lit/len: five waves, distances: one): every wave counts
// all chunks' lengths itself (ballots over S.lens: nothing to wait for), ranks its own chunk's symbols and stores them;
// chunk 0's wave also writes the per-length arrays.  The root tables are filled afterwards, one thread per ENTRY
// (par_fill).  (Rounds 2-3: one wave per alphabet, chunk after chunk, each symbol's lane filling its 2^(root-len) entries
// in a loop — 64 turns for a 4-bit code: 12k cycles on text with fourteen waves waiting.)
__device__ __forceinline__ static bool par_build(ParSmem& S, uint32_t base, uint32_t nsym, uint32_t chunk, bool is_dist, uint16_t* syms, uint32_t* first,
                                                 uint16_t* cnt, uint16_t* offs) {
  const uint32_t lane = zes_lane();
  const uint8_t* lens = S.lens + base;
  uint32_t c[16], before[16];
#pragma unroll
  for (int l = 0; l < 16; l++) c[l] = before[l] = 0;
  uint32_t myl = 0;
  for (uint32_t s0 = 0; s0 < nsym; s0 += 64) {
    const uint32_t s = s0 + lane;
    const uint32_t l = s < nsym ? lens[s] : 0u;
    if (s0 == 64u * chunk) {  // (uniform)
      myl = l;
#pragma unroll
      for (int k = 1; k < 16; k++) before[k] = c[k];
    }
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
  if (chunk == 0u) {
    if (!is_dist) {
      uint32_t n8 = 0;  // literals with an 8-bit code
      for (uint32_t s0 = 0; s0 < 256u; s0 += 64) n8 += (uint32_t)__popcll(__ballot(lens[s0 + lane] == 8u));
      if (lane == 0) {
        S.f8lo = fst[8];
        S.f8n = n8 >= 128u ? n8 : 0u;  // worth testing for only when most literals are 8 bits long
        S.f8off = ofs[8];
      }
    }
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
  }
  // rank of a symbol among the symbols of its length = those in earlier chunks + lower lanes of this chunk with the same
  // length (four ballots match the 4-bit length)
  uint64_t same = ~0ull;
#pragma unroll
  for (int bt = 0; bt < 4; bt++) {
    const bool bit = (myl >> bt) & 1u;
    const uint64_t bal = __ballot(bit);
    same &= bit ? bal : ~bal;
  }
  if (myl) {
    uint32_t o2 = 0;
#pragma unroll
    for (int k = 1; k < 16; k++)
      if ((int)myl == k) o2 = ofs[k] + before[k];
    syms[o2 + (uint32_t)__popcll(same & zes_lanemask_lt())] = (uint16_t)(64u * chunk + lane);
  }
  return true;
}
// One root-table entry: the symbol whose code is a prefix of the entry's index (bits in stream order), if one of at most
// ROOT bits is.  (The per-length arrays are read up front, all reads in flight; the symbol and the entry's fields once.)
template <uint32_t ROOT, bool IS_DIST>
__device__ __forceinline__ static void par_fill(uint32_t i, const uint16_t* syms, const uint32_t* first, const uint16_t* cnt, const uint16_t* offs,
                                                uint32_t* lut, uint8_t* blut) {
  uint32_t fst[ROOT + 1], cn[ROOT + 1], ofs[ROOT + 1];
#pragma unroll
  for (uint32_t l = 1; l <= ROOT; l++) {
    fst[l] = first[l];
    cn[l] = cnt[l];
    ofs[l] = offs[l];
  }
  const uint32_t msb = __brev(i) >> (32u - ROOT);  // the index as a code reads it
  uint32_t at = 0, len = 0;
#pragma unroll
  for (uint32_t l = 1; l <= ROOT; l++) {
    const uint32_t r = (msb >> (ROOT - l)) - fst[l];
    if (r < cn[l]) {
      at = ofs[l] + r;
      len = l;
    }
  }
  uint32_t ent = 0, bl = 0x80u;  // default: long code or no code -> the full tables decide
  if (len) {
    const uint32_t sym = syms[at];
    ent = IS_DIST ? entry_d(sym, len) : entry_l(sym, len);
    const uint32_t kind = (ent >> 8) & 3u;
    bl = (ent & 15u) + ((ent >> 4) & 15u);  // code bits + extra bits
    if (IS_DIST) bl = (kind == 2u) ? bl : 0x80u;
    else bl = (kind == 0u) ? bl : (kind == 2u) ? (bl | 0x40u) : 0x80u;  // EOB and 286/287 go the long way
  }
  lut[i] = ent;
  blut[i] = (uint8_t)bl;
}

// wave scans over 64 lanes (inclusive), DPP: four steps inside each row of 16, then the row totals carried over
template <int CTRL, int ROWS>
__device__ __forceinline__ static uint32_t ph_dpp(uint32_t old, uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)x, CTRL, ROWS, 0xf, false);
}
__device__ __forceinline__ static uint32_t ph_scan_add(uint32_t x) {
  x += ph_dpp<0x111, 0xf>(0u, x);
  x += ph_dpp<0x112, 0xf>(0u, x);
  x += ph_dpp<0x114, 0xf>(0u, x);
  x += ph_dpp<0x118, 0xf>(0u, x);
  x += ph_dpp<0x142, 0xa>(0u, x);  // row_bcast:15 into rows 1 and 3
  x += ph_dpp<0x143, 0xc>(0u, x);  // row_bcast:31 into rows 2 and 3
  return x;
}
__device__ __forceinline__ static uint32_t ph_scan_max(uint32_t x) {
  x = max(x, ph_dpp<0x111, 0xf>(0u, x));
  x = max(x, ph_dpp<0x112, 0xf>(0u, x));
  x = max(x, ph_dpp<0x114, 0xf>(0u, x));
  x = max(x, ph_dpp<0x118, 0xf>(0u, x));
  x = max(x, ph_dpp<0x142, 0xa>(0u, x));
  x = max(x, ph_dpp<0x143, 0xc>(0u, x));
  return x;
}

// ZES_DEBUG_PHASES: header-step stamps go to slots 16..21 of the block's row (pointer set by the kernel)
#define HSTAMP(i)                                                          \
  do {                                                                     \
    if (hdbg && zes_lane() == 0) hdbg[16 + (i)] = (unsigned long long)clock64(); \
  } while (0)
// Dynamic header, first part, by wave 0 (uniform): the fixed fields and the code-length code's table; leaves what the
// second part needs in S.h_*.  Returns false on anything T2/T3 should look at.
template <bool LDS>
__device__ __forceinline__ static bool par_header_fixed(ParSmem& S, const BitSrc& src, uint32_t limit, uint32_t start, unsigned long long* hdbg) {
  const uint32_t lane = zes_lane();
  HSTAMP(0);
  uint32_t bfinal, HLIT, HDIST, HCLEN;
  {
    const uint32_t i = start >> 5, sh = start & 31u;
    const uint64_t w = (uint64_t)src_ldw<LDS>(src, i) | ((uint64_t)src_ldw<LDS>(src, i + 1u) << 32);
    const uint32_t f = (uint32_t)(w >> sh);  // 32 bits from the block's first
    bfinal = f & 1u;
    if (((f >> 1) & 3u) != 2u) return false;
    HLIT = ((f >> 3) & 31u) + 257u;
    HDIST = ((f >> 8) & 31u) + 1u;
    HCLEN = ((f >> 13) & 15u) + 4u;
  }
  // lane k reads the k-th 3-bit length; lane s then takes the one of symbol s (kClOrder[k] == s)
  uint32_t mycl;
  {
    const uint32_t bp = start + 17u + 3u * (lane < 19u ? lane : 0u);
    const uint32_t j = bp >> 5, s2 = bp & 31u;
    const uint64_t w = (uint64_t)src_ldw<LDS>(src, j) | ((uint64_t)src_ldw<LDS>(src, j + 1u) << 32);
    const uint32_t v = lane < HCLEN ? ((uint32_t)(w >> s2) & 7u) : 0u;
    // position of symbol s in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 (src/const.ts:31-35), five bits each
    constexpr uint64_t inv_lo = 3ull | (17ull << 5) | (15ull << 10) | (13ull << 15) | (11ull << 20) | (9ull << 25) | (7ull << 30) | (5ull << 35) | (4ull << 40) |
                                (6ull << 45) | (8ull << 50) | (10ull << 55);                                                        // symbols 0..11
    constexpr uint64_t inv_hi = 12ull | (14ull << 5) | (16ull << 10) | (18ull << 15) | (0ull << 20) | (1ull << 25) | (2ull << 30);  // symbols 12..18
    const uint32_t from = lane < 12u ? (uint32_t)(inv_lo >> (5u * lane)) & 31u : lane < 19u ? (uint32_t)(inv_hi >> (5u * (lane - 12u))) & 31u : 63u;
    mycl = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from << 2), (int)v);
    if (lane >= 19u) mycl = 0;
  }
  // the 128-entry table, two entries per lane: the symbols go to LDS in (length, symbol) order, and an entry takes the one
  // whose code is a prefix of its index (bits in stream order).  (Before: every symbol's lane filled its 2^(7-len) entries in
  // a loop; the 1-bit and 2-bit codes of incompressible data's headers made that 64 + 32 turns.)
  uint32_t kraft = 0;
  {
    uint32_t code = 0, off = 0, e0 = 0, e1 = 0;
    const uint32_t i0 = __brev(lane) >> 25, i1 = __brev(lane + 64u) >> 25;  // the entries' indices as a code reads them
    uint8_t* sorted = reinterpret_cast<uint8_t*>(S.wave_sum);  // [19] (nobody else's until P2)
    static_assert(sizeof(S.wave_sum) >= 20, "room for the code-length code's symbols");
    uint32_t fst[8], cn[8], ofs[8];
#pragma unroll
    for (uint32_t l = 1; l <= 7; l++) {
      const bool mine = lane < 19 && mycl == l;
      const uint64_t m = __ballot(mine);
      const uint32_t n = (uint32_t)__popcll(m);
      if (mine) sorted[off + (uint32_t)__popcll(m & zes_lanemask_lt())] = (uint8_t)lane;
      fst[l] = code;
      cn[l] = n;
      ofs[l] = off;
      kraft += n << (7 - l);
      code = (code + n) << 1;
      off += n;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t a0 = 0, l0 = 0, a1 = 0, l1 = 0;  // where in sorted[], which length
#pragma unroll
    for (uint32_t l = 1; l <= 7; l++) {
      const uint32_t r0 = (i0 >> (7u - l)) - fst[l], r1 = (i1 >> (7u - l)) - fst[l];
      if (r0 < cn[l]) {
        a0 = ofs[l] + r0;
        l0 = l;
      }
      if (r1 < cn[l]) {
        a1 = ofs[l] + r1;
        l1 = l;
      }
    }
    e0 = l0 ? (uint32_t)sorted[a0] | (l0 << 5) : 0u;
    e1 = l1 ? (uint32_t)sorted[a1] | (l1 << 5) : 0u;
    if (kraft <= 128u) {  // (an over-subscribed code has no prefix property: declined below)
      S.cl_lut[lane] = (uint8_t)e0;
      S.cl_lut[lane + 64u] = (uint8_t)e1;
    }
  }
  if (kraft > 128u) return false;
  for (uint32_t i = lane; i < 352; i += 64) S.lens[i] = 0;
  if (lane == 0) {
    const uint32_t p = start + 17u + 3u * HCLEN;  // the first code-length symbol
    S.bfinal = bfinal;
    S.h_hlit = HLIT;
    S.h_total = HLIT + HDIST;
    S.h_wbase = p & ~31u;
    S.h_off = p & 31u;
    S.h_done = 0;
  }
  (void)limit;
  HSTAMP(1);
  return true;
}

// Dynamic header, second part, by ALL waves (uniform for the workgroup): the HLIT + HDIST code lengths into S.lens, the
// header's end into S.hdr_end; anything T2/T3 should look at sets S.status.  (src/inflate.ts:140-204.)
//   Rounds of 1024 bits, wave w on the 64 bit positions [64 w, 64 w + 64) of the round: lane j decodes the code-length symbol
//   that would start at bit j (one table lookup for all 64).  A symbol that starts in the window before can end at most 13
//   bits into this one, so the window has at most 14 entry offsets: pointer doubling (six lane permutes) gives every offset
//   the offset its chain leaves the window at, the code lengths the chain produces, whether it meets a hole of the code,
//   and the last plain length on it ("repeat previous" at the head of the next window takes that).  The waves' tables go
//   to LDS; every wave composes the windows before its own (<= 15 lookups) and knows its entry offset, the index of its
//   first length and the value in front — the rest is the lone wave's step of rounds 2-3 for all windows at once: the
//   real chain through the window is marked by a scalar walk, its symbols are moved into lanes 0.. in order, "repeat
//   previous" takes its value through a max-scan, a symbol's first entry is a prefix sum of the counts, and each lane
//   stores its own (at most six non-zero) entries.  (One wave, window after window: 27k cycles of the 60k a block's
//   header took, with fifteen waves waiting.)
template <bool LDS>
__device__ __forceinline__ static void par_header_lens(ParSmem& S, const BitSrc& src, uint32_t limit, unsigned long long* hdbg) {
  const uint32_t lane = zes_lane(), wave = threadIdx.x >> 6;
  constexpr uint32_t LP_NONE = 31u;
  uint32_t* s_tab = S.lut_d;             // [PAR_WAVES][32] (the first window of all is entered up to 31 bits in): the distance table is built later
  uint32_t* s_cl = S.lut_l + wave * 64u;  // this wave's 64 words: so is the lit/len table
  static_assert(PAR_WAVES * 32u <= (1u << PD_ROOT) && PAR_WAVES * 64u <= (1u << PL_ROOT), "header scratch lies over the tables");
  const uint32_t HLIT = S.h_hlit, total = S.h_total;
  uint32_t wbase0 = S.h_wbase, off0 = S.h_off, k0 = 0, prev0 = 0;
  for (;;) {  // (uniform for the workgroup)
    const uint32_t wb = wbase0 + 64u * wave;
    const uint32_t d = wb >> 5;
    const uint32_t w0 = src_ldw<LDS>(src, d), w1 = src_ldw<LDS>(src, d + 1), w2 = src_ldw<LDS>(src, d + 2);
    const uint64_t lo64 = (uint64_t)w0 | ((uint64_t)w1 << 32), hi64 = (uint64_t)w1 | ((uint64_t)w2 << 32);
    const uint32_t bits = lane < 32u ? (uint32_t)(lo64 >> lane) : (uint32_t)(hi64 >> (lane - 32u));  // >= 32 bits from bit `lane` on
    const uint32_t e = S.cl_lut[bits & 127u];
    const uint32_t l = e >> 5, sy = e & 31u;
    const uint32_t xb = sy == 16u ? 2u : sy == 17u ? 3u : sy == 18u ? 7u : 0u;
    const uint32_t xv = (bits >> l) & ((1u << xb) - 1u);
    const bool valid = l != 0u && (uint64_t)wb + lane + l + xb <= (uint64_t)limit;  // a code-length code matches here, inside the data
    const uint32_t rep = !valid ? 0u : sy == 16u ? 3u + xv : sy == 17u ? 3u + xv : sy == 18u ? 11u + xv : 1u;
    // packed per lane: [6:0] offset of the following symbol (up to 77; a hole: out of the window), [14:7] repeat count, [19:15] symbol, [20] valid
    const uint32_t pk = (valid ? lane + l + xb : 64u) | (rep << 7) | (sy << 15) | ((uint32_t)valid << 20);
    // the window's transfer table: [6:0] where the chain from here goes on (>= 64: out), [20:7] lengths it produces,
    // [21] it meets a hole, [26:22] the last plain length on it (LP_NONE: only "repeat previous" so far)
    uint32_t t = (pk & 127u) | (rep << 7) | ((uint32_t)!valid << 21) | ((!valid || sy == 16u ? LP_NONE : sy >= 17u ? 0u : sy) << 22);
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const uint32_t J = t & 127u;
      const uint32_t u = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(min(J, 63u) << 2), (int)t);
      if (J < 64u) {
        const uint32_t lpu = (u >> 22) & 31u;
        t = (u & 127u) | ((((t >> 7) & 0x3FFFu) + ((u >> 7) & 0x3FFFu)) << 7) | ((t | u) & (1u << 21)) | ((lpu != LP_NONE ? lpu : (t >> 22) & 31u) << 22);
      }
    }
    if (lane < 32u) s_tab[wave * 32u + lane] = t;
    __syncthreads();
    // this wave's entry: through the windows before it
    uint32_t en = off0, k = k0, prev = prev0;
    bool past = false, bad = false;
    for (uint32_t v = 0; v < wave; v++) {
      const uint32_t tt = s_tab[v * 32u + en];
      k += (tt >> 7) & 0x3FFFu;
      if (k >= total) {  // the sequence ends in window v: that wave's business
        past = true;
        break;
      }
      bad = bad || ((tt >> 21) & 1u);  // (all of window v's chain lies inside the sequence)
      const uint32_t lpv = (tt >> 22) & 31u;
      prev = lpv != LP_NONE ? lpv : prev;
      en = (tt & 127u) - 64u;
    }
    if (bad && lane == 0) atomicOr(&S.status, 1u);
    if (!past && !bad) {
      // the chain through the window: which lanes are on it
      uint64_t m = 0;
      uint32_t cur = en, nsym = 0;
      while (cur < 64u) {
        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)pk, (int)cur);
        m |= 1ull << cur;
        nsym++;
        cur = v & 127u;
      }
      // into lanes 0 .. nsym-1, in order
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if ((m >> lane) & 1ull) s_cl[(uint32_t)__popcll(m & zes_lanemask_lt())] = pk;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const uint32_t tc = s_cl[lane];
      const bool on = lane < nsym;
      const uint32_t s1 = on ? (tc >> 15) & 31u : 0u, r1 = on ? (tc >> 7) & 255u : 0u;
      const bool hole = on && !((tc >> 20) & 1u);
      const uint32_t is16 = (uint32_t)(on && !hole && s1 == 16u);
      const uint32_t valraw = s1 >= 16u ? 0u : s1;  // (17, 18: zeros; 16: the value of the symbol before, below)
      // the last symbol at or before this lane that is no "repeat previous" (0: none in this window — the carry)
      const uint32_t srcl = ph_scan_max((on && !is16) ? lane + 1u : 0u);
      const uint32_t vsrc = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((srcl ? srcl - 1u : 0u) << 2), (int)valraw);
      const uint32_t val = srcl ? vsrc : prev;
      const uint32_t kin = ph_scan_add(r1);
      const uint32_t kk = k + kin - r1;  // the symbol's first entry
      const uint64_t endm = __ballot(on && !hole && kk + r1 >= total);  // the symbol that completes the sequence (the first such)
      const uint32_t cut = endm ? (uint32_t)__builtin_ctzll(endm) : nsym - 1u;  // last symbol of this window that counts
      const bool mine = on && lane <= cut;
      const uint64_t badm = __ballot(mine && (hole || (is16 && kk == 0u) || kk + r1 > total));
      if (badm) {
        if (lane == 0) atomicOr(&S.status, 1u);
      } else {
        if (mine && val) {
          for (uint32_t q = 0; q < r1; q++) {  // (a non-zero value repeats at most six times)
            const uint32_t idx = kk + q;
            S.lens[idx < HLIT ? idx : 288 + (idx - HLIT)] = (uint8_t)val;
          }
        }
        if (endm) {  // the header ends behind that symbol
          const uint32_t nx = (uint32_t)__builtin_amdgcn_readlane((int)(tc & 127u), (int)cut);
          if (lane == 0) {
            S.hdr_end = wb + nx;
            S.h_done = 1;
          }
        } else if (wave == PAR_WAVES - 1u && lane == 0) {  // the next round's entry
          S.h_off = cur - 64u;
          S.h_k = (uint32_t)__builtin_amdgcn_readlane((int)(kk + r1), (int)cut);
          S.h_prev = (uint32_t)__builtin_amdgcn_readlane((int)val, (int)cut);
        }
      }
    }
    __syncthreads();
    if (S.status || S.h_done) break;
    off0 = S.h_off;
    k0 = S.h_k;
    prev0 = S.h_prev;
    wbase0 += 64u * PAR_WAVES;
    __syncthreads();  // (the carry is read before the next round's last wave writes it again)
  }
  if (wave == 0) HSTAMP(2);
  if (!S.status && S.hdr_end > limit && threadIdx.x == 0) S.status = 1;
}

// What a work item is and where its results go.  T1 (k_inf_block_par): candidate ci of a reference-made stream, output
// slot w of 131072 bytes, a ZesCandRes.  T2 (k_inf_seg_block_par, FOREIGN): one block of any encoder's stream — a
// block may copy from the 32 KiB in front of it and has any length up to the LDS image — whose output is 16-bit
// symbols in the segment-parallel tier's symbol store (a byte, or 256 + index into the window in front of the block),
// the map of its last 32 Ki symbols and a ZesSegRes: the same products as the wave decoder's (k_inf_seg_scan), which
// still takes what this kernel declines (stored and fixed blocks, blocks behind an unlisted start, > 128 KiB).
struct ParItem {
  const uint32_t* g32;  // the buffer as dwords
  uint32_t lastdw, limit, start;
  uint32_t de_est, de_est2;  // end estimates: the next listed block start, the one after it (0: none)
  // T1
  ZesCandRes* cres;  // this item's result
  ZesCandRes* cres_host;  // (one-buffer calls) a copy of it in the host's page-locked memory, or null ...
  uint32_t* start_host;   // ... and where the block's start bit goes
  uint8_t* dst;      // its output slot
  uint64_t room;     // bytes that may be stored there
  // T2
  ZesSegRes* sres;
  uint32_t* sym;      // the item's share of the symbol store (two symbols per dword)
  uint64_t sym_cap;   // symbols it holds
  uint32_t* sym_all;  // the store; behind the shares: a common area handed out by need (a block whose share a false
  unsigned long long* bump;  // candidate has cut short, or that inflates further than its share)
  uint64_t bump_base, bump_cap;  // first dword of that area, symbols it holds
  uint64_t* symoff;   // out: where the item's symbols are (dword offset into the store)
  uint32_t* map;      // [ZES_WINDOW / 2]
  const uint32_t* cand;
  uint32_t ncand;
  uint32_t* fail_list;  // items left to the wave decoder: [0] = count, then the items
  uint32_t w;
};

// TWO: the transfer tables are built over two windows (compressible data: long segments whose trajectories merge
// behind the first window); incompressible data takes the 8-bit-literal construction either way and runs ~4 %
// faster in the smaller kernel, so the host picks the variant by the stream's size against its output's.
template <bool FOREIGN, bool TWO>
__device__ __forceinline__ static void par_body(ParSmem& S, const ParItem& it, unsigned long long* dbg) {
#define STAMP(i)                                                     \
  do {                                                               \
    if (dbg && threadIdx.x == 0) dbg[(size_t)blockIdx.x * ZES_PAR_DBG_ROW + (i)] = (unsigned long long)clock64(); \
  } while (0)
  // the item could not be decoded here: T1 reports it (the chain check decides), T2 hands it to the wave decoder
#define PAR_DECLINE(END_BIT, TOTAL)                                   \
  do {                                                                \
    if (threadIdx.x == 0) {                                           \
      if (FOREIGN) {                                                  \
        ZesSegRes r_;                                                 \
        r_.end_bit = 0;                                               \
        r_.out_len = 0;                                               \
        r_.flags = 0;                                                 \
        r_.next = 0;                                                  \
        *it.sres = r_;                                                \
        it.fail_list[1u + atomicAdd(&it.fail_list[0], 1u)] = it.w;    \
      } else {                                                        \
        ZesCandRes r_;                                                \
        r_.end_bit = (END_BIT);                                       \
        r_.out_len = (TOTAL);                                         \
        r_.flags = 0;                                                 \
        *it.cres = r_;                                                \
        if (it.cres_host) {                                           \
          *it.cres_host = r_;                                         \
          *it.start_host = it.start;                                  \
        }                                                             \
      }                                                               \
    }                                                                 \
  } while (0)
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t lastdw = it.lastdw, limit = it.limit, start = it.start;
  BitSrc src;
  src.g32 = it.g32;
  src.lastdw = lastdw;
  src.s32 = reinterpret_cast<const uint32_t*>(S.out);
  src.s_first = start >> 5;
  uint32_t ds = 0, seg0 = 0, seglen = 0, base = 0, stop = 0, ecode = 0, tail_code = 0;  // (seg0: first bit of segment 0)
  uint32_t plimit = limit;  // end of what the phases that read the staged copy may look at: the data's end, or the staged copy's
  Lit8 f8 = {0, 0, 0};
  bool p2_global = false;  // the count pass reads its bits from global memory (the tables were parked over the staged block)
  STAMP(0);
  // P0 + P1 for one estimate of the block's end.  Returns 0 = go on, 1 = not decodable here.
  auto stage_and_tables = [&](const uint32_t de_est) __attribute__((always_inline)) -> uint32_t {
    // ---- P0: stage the block's compressed bytes in LDS (swizzled), header + tables by wave 0 ----
    {
      // segment rounding + one token + slack behind the estimate; the longest header (17 + 57 + 320 x 14 bits) in any case
      const uint64_t end_bit = min((uint64_t)limit, max((uint64_t)de_est, (uint64_t)start + 4800u) + 1280u);
      const uint32_t end_dw = min(lastdw, (uint32_t)((end_bit + 63u) >> 5));
      src.s_count = end_dw - src.s_first + 1u;
      // Segments are at least 64 bits long, so for a short estimate (a block of less than 8 KiB, or a false candidate
      // right behind the block's start) the 1024 segments — and even the header — reach beyond the staged bytes.  A
      // chain that gets there must not go on through whatever the staging area holds: behind `plimit` every position
      // fails, the item is declined, and the wave decoder (T2) or the chain check (T1) has it.  (tools/gpu_fuzz.py seed
      // 910: a damaged block ran past both estimates, walked stale bytes of the area to the last lane, and the serial
      // tail from there met a real block end — garbage accepted where the reference throws.)
      plimit = (uint32_t)min((uint64_t)limit, ((uint64_t)end_dw + 1u) * 32u);
    }
    // a block whose compressed bytes do not fit the staging area (> 144 KiB for <= 128 KiB of output)
    // is not reference-made: leave it to T2
    const bool use_lds = src.s_count <= STAGE_DW;
    if (tid == 0) S.status = use_lds ? 0u : 1u;
    if (use_lds) {
      uint32_t* st = reinterpret_cast<uint32_t*>(S.out);
      for (uint32_t k = tid; k < src.s_count; k += PAR_THREADS) st[k ^ ((k >> 5) & 31u)] = src.g32[src.s_first + k];
    }
    __syncthreads();
    unsigned long long* hdbg = dbg ? dbg + (size_t)blockIdx.x * ZES_PAR_DBG_ROW : nullptr;
    if (wave == 0 && use_lds) {
      const bool ok = par_header_fixed<true>(S, src, plimit, start, hdbg);
      if (!ok && lane == 0) S.status = 1;
    }
    __syncthreads();
    if (S.status) return 1u;
    par_header_lens<true>(S, src, plimit, hdbg);
    __syncthreads();
    if (S.status) return 1u;
    // the two alphabets' canonical arrays, a wave per 64 symbols; then the root tables, a thread per entry
    if (wave < 5u) {
      if (!par_build(S, 0, 288, wave, false, S.syms_l, S.first_l, S.cnt_l, S.offs_l) && lane == 0) atomicOr(&S.status, 1u);
    } else if (wave == 5u) {
      if (!par_build(S, 288, 32, 0, true, S.syms_d, S.first_d, S.cnt_d, S.offs_d) && lane == 0) atomicOr(&S.status, 1u);
    }
    __syncthreads();
    if (wave == 0) HSTAMP(3);
    if (S.status) return 1u;
    static_assert(PAR_THREADS == (1u << PL_ROOT) && PAR_THREADS >= (1u << PD_ROOT), "one thread per root-table entry");
    par_fill<PL_ROOT, false>(tid, S.syms_l, S.first_l, S.cnt_l, S.offs_l, S.lut_l, S.len_l);
    if (tid < (1u << PD_ROOT)) par_fill<PD_ROOT, true>(tid, S.syms_d, S.first_d, S.cnt_d, S.offs_d, S.lut_d, S.len_d);
    __syncthreads();
    if (wave == 0) HSTAMP(4);
    STAMP(1);
    ds = S.hdr_end;
    // The reference's match finder never looks in front of the block (src/lz77.ts:11-22: the index is built per block),
    // so its blocks begin with a literal and an early match reaches back no further than the bytes before it.  Another
    // encoder's blocks — which pass this tier's header rules more often than not — nearly always begin with matches
    // into the block before.  The first wavefront decodes the token at each of the block's first 64 bit positions,
    // walks the chain of real tokens through them (~10 of them) and declines the block here, in front of the table
    // construction and the count pass, when one reaches in front of the block (before: all blocks of such a stream
    // were decoded side by side up to the count pass, where their sizes gave them away: 0.48 of the 1.34 ms of 16 MiB of
    // zlib -9 text, 1.9 of the 8.0 ms of 256 x 1 MiB).  (A lone lane decoding them one after the other: 15k cycles.)
    // (The form of the decoder for compressible data only: incompressible data from another encoder comes in stored
    // blocks, and the other form's blocks — 0.55 ms per 64 MiB — would pay 2 % for the test.)
    if (!FOREIGN && TWO) {
      if (wave == 0) {
        LaneBits pb;
        lb_seek<true>(pb, src, ds + lane);
        uint32_t v = 0, len = 0, dist = 0;
        const uint32_t kind = (ds + lane < plimit) ? tok_step<true>(S, pb, src, v, len, dist) : T_FAIL;
        const uint32_t w0 = kind | ((pb.pos - (ds + lane)) << 2) | (len << 8);  // kind, bits of the token (<= 48), match length
        uint32_t cur = 0, made = 0;
        bool hist = false;
        for (uint32_t k = 0; k < 16u && cur < 64u; k++) {
          const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)w0, (int)cur);
          const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)dist, (int)cur);
          const uint32_t kd = a & 3u;
          if (kd == T_LIT) {
            made++;
          } else if (kd == T_MATCH) {
            if (d > made) {
              hist = true;
              break;
            }
            made += (a >> 8) & 511u;
          } else {
            break;  // (end of block, or a bad code: the passes below deal with it)
          }
          cur += (a >> 2) & 63u;
        }
        if (hist && lane == 0) S.status = 1u;
      }
      __syncthreads();
      if (S.status) return 1u;
    }
    f8.lo = S.f8lo;
    f8.n = S.f8n;
    f8.off = S.f8off;
    const uint32_t tab_off = ((src.s_count + 31u) & ~31u) * 4u;  // the swizzle permutes inside rows of 32 dwords
    {
      const uint32_t span = de_est > ds ? de_est - ds : 1u;
      seglen = max(64u, (span + PAR_THREADS - 1) / PAR_THREADS);
    }
    // compressible data with room behind the staged block: every bit position's exit code, last first (seg_table_dp);
    // its segments are whole dwords, so segment 0 starts up to 31 bits in front of the header's end
    const bool dp = TWO && tab_off + DP_BYTES <= STAGE_DW * 4u && !(f8.n && seglen >= 256u);  // (uniform)
    seg0 = ds;
    if (dp) {
      seg0 = ds & ~31u;
      const uint32_t span = de_est > seg0 ? de_est - seg0 : 1u;
      seglen = (max(64u, (span + PAR_THREADS - 1) / PAR_THREADS) + 31u) & ~31u;
    }

    // ---- P1: transfer tables in registers, then composition by lane broadcasts ----
    const uint64_t b_me64 = (uint64_t)seg0 + (uint64_t)tid * seglen;
    base = (uint32_t)(b_me64 < 0xFFFFFF00ull ? b_me64 : 0xFFFFFF00ull);
    stop = (uint32_t)((b_me64 + seglen) < 0xFFFFFF00ull ? (b_me64 + seglen) : 0xFFFFFF00ull);
    SegTab tab = {0, 0, 0, 0, 0, 0, 0, 0};
    if (dp) {
      uint8_t* T13 = S.out + tab_off + PAR_THREADS * DP_ROW;
      uint8_t* D12 = T13 + (1u << DP_LROOT);
      dp_build(S, T13, D12);
      __syncthreads();
      seg_table_dp(S, src, plimit, base, seglen, T13, D12, S.out + tab_off + tid * DP_ROW);
    } else {
      seg_table<true, TWO>(S, src, plimit, base, stop, f8, tab, dbg ? dbg + (size_t)blockIdx.x * ZES_PAR_DBG_ROW : nullptr);
    }
    STAMP(2);
    // Composition.  With room behind the staged block (compressible data: the block's bytes fill a third of the
    // staging area) every lane parks its table in LDS, 52 bytes apart (the sweep's tables are there already, 68 apart),
    // and a step of the walks below is one byte read; otherwise (incompressible data) the tables stay in registers and
    // a step broadcasts a lane's table.
    // Round 4: incompressible data has no room behind its staged block (131 KB of the 144 KB area) and kept the tables in
    // registers — a step of the walks broadcast a lane's table, sixteen readlanes and a select chain: 155k of the kernel's
    // 450k cycles per block of random data.  The tables are parked OVER the staged block instead: once every lane has built
    // its table nothing reads the staged copy any more but the count pass, which then takes its bits from global memory
    // like the emit pass does anyway.
    const bool room = dp || tab_off + PAR_THREADS * 52u <= STAGE_DW * 4u;  // uniform
    const bool over = !room && !FOREIGN;
    const bool lds_tabs = room || over;
    const uint32_t toff = over ? 0u : tab_off;
    p2_global = over;
    const uint32_t tstride = dp ? DP_ROW : 52u;
    const uint8_t* tb = S.out + toff + wave * 64u * tstride;  // this wave's 64 tables
    if (over) __syncthreads();  // every lane is done with the staged copy
    if (dp) {
      __syncthreads();
    } else if (lds_tabs) {
      uint32_t* tw = reinterpret_cast<uint32_t*>(S.out + toff) + tid * 13u;
      tw[0] = (uint32_t)tab.a;
      tw[1] = (uint32_t)(tab.a >> 32);
      tw[2] = (uint32_t)tab.b;
      tw[3] = (uint32_t)(tab.b >> 32);
      tw[4] = (uint32_t)tab.c;
      tw[5] = (uint32_t)(tab.c >> 32);
      tw[6] = (uint32_t)tab.d;
      tw[7] = (uint32_t)(tab.d >> 32);
      tw[8] = (uint32_t)tab.e;
      tw[9] = (uint32_t)(tab.e >> 32);
      tw[10] = (uint32_t)tab.f;
      tw[11] = (uint32_t)(tab.f >> 32);
      __syncthreads();
    }
    {
      // over the 64 segments of this wave: lane j (< 48) carries input offset j
      uint32_t cur = lane;
      if (lds_tabs) {
        // A walk through 64 tables is 64 dependent LDS reads, ~350 cycles each with sixteen waves at it: the two walks of
        // this step were 45k of a block's cycles (text), 60k (incompressible data).  In four groups of 16 segments: the
        // 4 x 48 group tables first (three chains per lane, side by side: 16 reads deep), then the wave's table through
        // the four of them; the true entries below walk the groups side by side as well.  20 + 20 reads deep, not 128.
        uint8_t* gt = &S.gtab[wave][0][0];
#pragma unroll
        for (uint32_t q = 0; q < 3u; q++) {
          const uint32_t idx = lane + 64u * q, grp = idx / 48u, j = idx - 48u * grp;  // (idx < 192)
          uint32_t c = j;
          for (uint32_t sgm = 0; sgm < 16u; sgm++) {
            const uint32_t v = tb[(16u * grp + sgm) * tstride + (c < 48u ? c : 0u)];
            c = c < 48u ? v : c;  // (end of block / fail stay what they are)
          }
          gt[idx] = (uint8_t)c;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's own writes, read back by its other lanes)
#pragma unroll
        for (uint32_t grp = 0; grp < 4u; grp++) {
          const uint32_t v = gt[48u * grp + (cur < 48u ? cur : 0u)];
          cur = cur < 48u ? v : cur;
        }
      } else {
        for (uint32_t sgm = 0; sgm < 64u; sgm++) {
          const SegTab ws = tab_bcast(tab, sgm);
          if (cur < 48u) cur = tab_get(ws, cur);
        }
      }
      if (lane < 48u) S.wtab[wave][lane] = (uint8_t)cur;
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t e = ds - seg0;  // the first token starts at the header's end
      for (uint32_t k = 0; k < PAR_WAVES; k++) {
        S.wentry[k] = (uint8_t)e;
        if (e < 48u) e = S.wtab[k][e];
      }
      S.tail_entry = e;  // state after the last segment: offset past its end, or EOB / fail
    }
    __syncthreads();
    // the wave's true entry walks through its 64 segments
    uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.wentry[wave]);
    uint32_t mine = C_FAIL;
    if (lds_tabs) {
      // the entries of the four groups (three lookups), then every lane walks its own group up to its own segment
      const uint8_t* gt = &S.gtab[wave][0][0];
      const uint32_t grp = lane >> 4, upto = lane & 15u;
      uint32_t eg = e, c = e;
#pragma unroll
      for (uint32_t k = 0; k < 3u; k++) {
        eg = eg < 48u ? (uint32_t)gt[48u * k + eg] : eg;
        c = grp == k + 1u ? eg : c;
      }
      for (uint32_t sgm = 0; sgm < 15u; sgm++) {
        const uint32_t v = tb[(16u * grp + sgm) * tstride + (c < 48u ? c : 0u)];
        c = (sgm < upto && c < 48u) ? v : c;
      }
      mine = c;
    } else {
      // every lane looks the (uniform) entry up in its own table and the owning lane's answer is broadcast
      for (uint32_t sgm = 0; sgm < 64u; sgm++) {
        if (lane == sgm) mine = e;
        const uint32_t own = e < 48u ? tab_get(tab, e) : e;
        e = (uint32_t)__builtin_amdgcn_readlane((int)own, (int)sgm);
      }
    }
    ecode = mine;
    tail_code = S.tail_entry;
    return 0u;
  };
  // estimate of the block's end: the next candidate on the list (exact on a clean chain)
  uint32_t bad = stage_and_tables(it.de_est);
  // The chain is still alive after the last segment: the estimate was a false candidate inside
  // this block.  Take the candidate after it as the estimate and decode once more in parallel (a
  // serial tail from here can cost ~10 ms); a second false candidate in the same block falls to
  // the serial tail below.  Written as a second straight-line copy, not a loop: a back edge makes
  // the compiler hoist invariants across the whole decoder and spill.
  // (Round 3: also when the chain FAILED on the first estimate.  Since the staged copy's end bounds the phases that
  // read it, a false candidate a little behind the block's start makes the chain fail at that end rather than run on:
  // the item used to be declined there — a whole block left to a lone wave of the wave decoder, 7 ms per 100 KiB, e.g.
  // 8 MiB of zlib text in 8.1 ms instead of 1.3.  A block that is really damaged fails a second time and is declined.)
  if (!bad && (tail_code < 48u || tail_code == C_FAIL) && it.de_est2) {
    __syncthreads();
    bad = stage_and_tables(it.de_est2);
  }
  if (bad) {
    PAR_DECLINE(start, 0u);
    return;
  }
  STAMP(3);

  // ---- P2: count pass from the true entries, totals, end bit, output offsets ----
  uint32_t entry = base + ecode, exit_pos = 0, outbytes = 0, flags = 0;
  if (!FOREIGN && p2_global)  // (uniform; the same bound as the staged form: behind plimit every position fails)
    seg_run<false, false, FOREIGN>(S, src, plimit, entry, stop, ecode < 48u, f8, 0, exit_pos, outbytes, flags);
  else
    seg_run<false, true, FOREIGN>(S, src, plimit, entry, stop, ecode < 48u, f8, 0, exit_pos, outbytes, flags);
  STAMP(22);  // (ZES_DEBUG_PHASES: wave 0 is through the count pass)
  if (dbg && threadIdx.x == PAR_THREADS - 64) dbg[(size_t)blockIdx.x * ZES_PAR_DBG_ROW + 23] = (unsigned long long)clock64();
  if (ecode >= 48u) flags = F_VOID;
  if (tid == 0) {
    S.tail_bytes = 0;
    S.tail_end = 0;
  }
  __syncthreads();  // every lane is done with the staged copy: the image and the bitmap are free
  for (uint32_t i = tid; i < ZES_BLK / 32; i += PAR_THREADS) S.bitmap[i] = 0;
  // a chain that is still alive after the last segment (the end estimate was short: a false
  // candidate sits inside this block) is finished serially by one lane, from global memory
  uint32_t tail_start = 0;
  if (tail_code < 48u) {
    const uint64_t last_stop = (uint64_t)seg0 + (uint64_t)PAR_THREADS * seglen;
    tail_start = (uint32_t)(last_stop + tail_code);
    if (tid == 0) {
      uint32_t ex, ob, fl;
      seg_decode<false, false, FOREIGN>(S, src, limit, tail_start, 0xFFFFFF00u, 0, ex, ob, fl);
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
    // (T1: a block of a reference-made stream holds exactly 131072 bytes unless it is the last one — anything else is a
    // false candidate or another encoder's block, and the chain check will refuse it: no emit pass for it.  A batch of
    // 256 zlib streams paid 2.3 ms for this tier's attempt.)
    const bool good = (st & 4u) && !(st & 2u) && total <= (FOREIGN ? PAR_MAX_OUT : ZES_BLK) && (FOREIGN || S.bfinal || total == ZES_BLK);
    if (!good) {
      PAR_DECLINE(end_bit, total);
      return;
    }
    const uint32_t my_off = wbase + incl - outbytes;
    STAMP(4);
    // T2: where the symbols go — the item's share of the store, or, when that is too small, a piece of the common area
    uint32_t* symp = it.sym;
    if (FOREIGN) {
      if ((uint64_t)total > it.sym_cap) {
        if (tid == 0) {
          const unsigned long long o = atomicAdd(it.bump, (unsigned long long)((total + 7u) & ~7u));
          S.sp_n = (o + total <= it.bump_cap) ? (uint32_t)(o >> 3) : 0xFFFFFFFFu;  // (units of 8 symbols)
        }
        __syncthreads();
        const uint32_t o8 = S.sp_n;
        __syncthreads();
        if (o8 == 0xFFFFFFFFu) {
          PAR_DECLINE(end_bit, total);
          return;
        }
        symp = it.sym_all + it.bump_base + (uint64_t)o8 * 4u;
      }
      if (tid == 0) *it.symoff = (uint64_t)(symp - it.sym_all);
    }

    // T2: a block longer than the image is emitted and resolved chunk by chunk (clo = first byte of the chunk); T1 and
    // blocks of up to PAR_CHUNK bytes make one pass.  (A jump back instead of a loop: T1 gets no back edge.)
    uint32_t clo = 0;
    static_assert(PAR_CHUNK % RES_W == 0 && PAR_DIST_OFF + 2u * RES_W <= ZES_BLK, "T2: the distance array lies behind the chunk");
  next_chunk:
    const uint32_t clen = FOREIGN ? min(PAR_CHUNK, total - clo) : total;
    if (FOREIGN) {
      if (tid < 4u) S.res_strad[tid] = 0u;  // (the emit pass sets the match that runs in from the chunk before)
      if (clo) {
        for (uint32_t i = tid; i < ZES_BLK / 32; i += PAR_THREADS) S.bitmap[i] = 0;
      }
      __syncthreads();
    }
    // ---- P3: emit (compressed bits from global memory: the LDS now holds the output image) ----
    uint32_t f2 = 0;
    {
      uint32_t ex2, ob2;
      const Lit8 f8e = (FOREIGN && total > PAR_CHUNK) ? Lit8{0, 0, 0} : f8;  // (the four-literal store is not clipped to a chunk)
      // T2: a lane whose output lies outside the chunk has nothing to emit
      const bool inchunk = !FOREIGN || (my_off < clo + PAR_CHUNK && my_off + outbytes > clo);
      seg_run<true, false, FOREIGN>(S, src, limit, entry, stop, !(flags & F_VOID) && entry < stop && inchunk, f8e, my_off, ex2, ob2, f2, clo);
    }
    if (tid == 0 && tail_code < 48u) {
      uint32_t ex2, ob2, f3 = 0;
      seg_decode<true, false, FOREIGN>(S, src, limit, tail_start, 0xFFFFFF00u, seg_total, ex2, ob2, f3, clo);
      f2 |= f3;
    }
    if (f2 & F_HIST) atomicOr(&S.status, 8u);
    __syncthreads();
    if (S.status & 8u) {
      PAR_DECLINE(end_bit, total);
      return;
    }

    STAMP(5);
    // ---- P4: match resolution by all 16 waves, window by window (src/inflate.ts:287-290).
    // A copy may read bytes that an earlier copy produced, and on text such chains are hundreds of copies deep
    // (every phrase copies its nearest earlier occurrence), so neither "in order, 64 at a time" (one wave: 40 %
    // of this kernel on text) nor "everything that is ready, round after round" (hundreds of rounds) scales.
    // Per BYTE the structure is a forest: byte b of a match has the parent b - D, a literal byte is a root.  For
    // one window of RES_W bytes at a time (everything below the window is final) every match byte gets its
    // distance to an ancestor in dist[] (u16: < RES_W + 32768), pointer jumping dist[b] += dist[b - dist[b]]
    // halves the depth per round until every pointer ends on a final byte — a literal of the window or a byte
    // below it — and then all bytes of the window are copied at once.  Overlapping copies (distance < length)
    // need no special case: byte p+5 of a run with D = 1 simply has the parent p+4.
    {
      // T1: the decode tables are dead, [RES_W] entries over them; T2: behind the chunk (the tables serve the next one)
      uint16_t* dist = FOREIGN ? reinterpret_cast<uint16_t*>(S.out + PAR_DIST_OFF) : reinterpret_cast<uint16_t*>(S.lut_l);
      if (tid < 3u) S.res_flag[tid] = 0u;
      if (!FOREIGN && tid < 4u) S.res_strad[tid] = 0u;
      __syncthreads();
      unsigned long long tacc[4] = {0, 0, 0, 0}, tlast = dbg ? clock64() : 0ull, nrounds = 0;  // ZES_DEBUG_PHASES: cycles of the four steps
#define P4LAP(i)                                         \
  do {                                                   \
    if (dbg && tid == 0) {                               \
      const unsigned long long now_ = clock64();         \
      tacc[i] += now_ - tlast;                           \
      tlast = now_;                                      \
    }                                                    \
  } while (0)
      // T1, few matches in the whole block (incompressible data: ~260 of 3-4 bytes among 131072 literals): the windows
      // below cost ~4.5k cycles each whatever is in them — 140k cycles a block.  Instead: list the matches (positions
      // ascending), find the ones whose source bytes are all literals — nearly all: they copy from final bytes, side by
      // side, one wave per match — and let one wave do the others in order.  Byte j of a match is byte (j mod D) of the
      // D bytes in front of it, overlapping or not (src/inflate.ts:287-290 copies byte by byte).
      bool sparse_done = false;
      uint32_t nstarts = 0;  // T1: matches in the block
      if (!FOREIGN) {
        static_assert(sizeof(S.lut_l) >= P4_SPARSE_MAX * 4u && sizeof(S.lut_d) >= P4_SPARSE_MAX, "the lists lie over the dead decode tables");
        uint32_t* sp_list = S.lut_l;
        uint8_t* sp_dep = reinterpret_cast<uint8_t*>(S.lut_d);
        uint32_t bw[ZES_BLK / 32 / PAR_THREADS], cnt4 = 0;
#pragma unroll
        for (uint32_t k = 0; k < ZES_BLK / 32 / PAR_THREADS; k++) {
          bw[k] = S.bitmap[tid * (ZES_BLK / 32 / PAR_THREADS) + k];
          cnt4 += (uint32_t)__popc(bw[k]);
        }
        uint32_t incl = cnt4;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
          const uint32_t t = __shfl_up(incl, dlt);
          if ((int)lane >= dlt) incl += t;
        }
        __syncthreads();  // (wave_sum is read by the phases before)
        if (lane == 63) S.wave_sum[wave] = incl;
        __syncthreads();
        uint32_t wbase2 = 0, nm = 0;
#pragma unroll
        for (uint32_t k = 0; k < PAR_WAVES; k++) {
          const uint32_t sm = S.wave_sum[k];
          if (k < wave) wbase2 += sm;
          nm += sm;
        }
        nstarts = nm;
        auto rec_of = [&](uint32_t P, uint32_t& D, uint32_t& L) {
          D = ((uint32_t)S.out[P] | ((uint32_t)S.out[P + 1u] << 8)) + 1u;
          L = (uint32_t)S.out[P + 2u] + 3u;
        };
        if (tid == 0) S.sp_n = 0;
        if (nm <= P4_SPARSE_MAX) {  // (uniform)
          uint32_t kk = wbase2 + incl - cnt4;
#pragma unroll
          for (uint32_t k = 0; k < ZES_BLK / 32 / PAR_THREADS; k++) {
            uint32_t w = bw[k];
            while (w) {
              const uint32_t bit = (uint32_t)__builtin_ctz(w);
              w &= w - 1u;
              sp_list[kk++] = 32u * (tid * (ZES_BLK / 32 / PAR_THREADS) + k) + bit;
            }
          }
          __syncthreads();
          // which matches copy bytes that another match produces?  (the list is sorted: lower bound, then a short walk)
          if (tid < nm) {
            uint32_t P = sp_list[tid], D, L;
            rec_of(P, D, L);
            const uint32_t a = P - D, b = a + min(D, L);  // source bytes [a, b)   (D <= P in this tier: checked at emit)
            const uint32_t from = a > (ZES_MAXMATCH - 1u) ? a - (ZES_MAXMATCH - 1u) : 0u;
            uint32_t lo = 0, hi = nm;
            while (lo < hi) {
              const uint32_t mid = (lo + hi) >> 1;
              if (sp_list[mid] < from) lo = mid + 1u; else hi = mid;
            }
            uint32_t dep = 0;
            for (; lo < nm; lo++) {
              const uint32_t Q = sp_list[lo];
              if (Q >= b) break;
              if (Q + (uint32_t)S.out[Q + 2u] + 3u > a) {
                dep = 1;
                break;
              }
            }
            sp_dep[tid] = (uint8_t)dep;
            if (dep) atomicAdd(&S.sp_n, 1u);
          }
          __syncthreads();
          // many of them (the 4 KiB pattern: every match copies what the match 4096 bytes earlier produced): one wave doing
          // them in order would be slower than the windows below (measured 561k cycles against 230k)
          sparse_done = S.sp_n <= P4_SPARSE_DEP_MAX;
        }
        if (sparse_done) {
          auto copy_match = [&](uint32_t P, uint32_t L, uint32_t D) {
            // byte j <- byte (j mod D) of the D bytes in front of P; j / D by a 32-bit reciprocal (j < 512: exact)
            const uint32_t rcp = (D > 1u && D < L) ? 0xFFFFFFFFu / D + 1u : 0u;
            for (uint32_t j = lane; j < L; j += 64u) {
              const uint32_t q = D >= L ? 0u : (D > 1u ? __umulhi(j, rcp) : j);
              S.out[P + j] = S.out[P - D + (j - q * D)];
            }
          };
          for (uint32_t i = wave; i < nm; i += PAR_WAVES) {
            if (sp_dep[i]) continue;
            uint32_t P = sp_list[i], D, L;
            rec_of(P, D, L);
            copy_match(P, L, D);
          }
          __syncthreads();
          if (wave == 0u) {
            for (uint32_t base = 0; base < nm; base += 64u) {
              uint64_t m = __ballot(base + lane < nm && sp_dep[base + lane] != 0u);
              while (m) {
                const uint32_t i = base + (uint32_t)__builtin_ctzll(m);
                m &= m - 1ull;
                uint32_t P = sp_list[i], D, L;
                rec_of(P, D, L);
                copy_match(P, L, D);
              }
            }
          }
          __syncthreads();
        }
      }
      // long matches (periodic data: 258 bytes each, a start every eighth word of the bitmap): most bytes would look four
      // to nine words back for their match — the running maxima serve them in one lookup (measured: fill 197k against
      // 70k cycles a block); text, a start every few bytes, finds it in the byte's own word or the one before
      if (FOREIGN) {  // (T1 has counted its matches for the sparse form above)
        uint32_t c4 = 0;
        for (uint32_t i = tid; i < (clen + 31u) / 32u; i += PAR_THREADS) c4 += (uint32_t)__popc(S.bitmap[i]);
        if (tid == 0) S.sp_n = 0;
        __syncthreads();
        for (int dlt = 32; dlt >= 1; dlt >>= 1) c4 += (uint32_t)__shfl_xor((int)c4, dlt);
        if (lane == 0 && c4) atomicAdd(&S.sp_n, c4);
        __syncthreads();
        nstarts = S.sp_n;
      }
      const bool scanmode = (uint64_t)nstarts * 48u < clen;  // (uniform)
      for (uint32_t ws = 0; ws < (sparse_done ? 0u : clen); ws += RES_W) {
        const uint32_t wlen = min(RES_W, clen - ws);
        // (a) per 32 positions of the window: the last match start at or before them (matches do not overlap each
        // other, so the nearest start in front of a byte is the only match that can cover it)
        // (T1 looks the start up in the bitmap itself, below: a match is at most 258 bytes long, so the start that covers a
        // byte is in its own word of the bitmap or one of the nine before it — no running maxima, no barrier, and no
        // carry from window to window.  T2 overwrites the words of resolved windows with its marker flags.)
        if (scanmode) {
          if (tid < RES_W / 32u) {
            const uint32_t w = S.bitmap[(ws >> 5) + tid];
            uint32_t hs = w ? 32u * tid + 32u - (uint32_t)__clz(w) : 0u;  // start + 1 (window-relative), 0: none
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) {
              const uint32_t t = __shfl_up(hs, dlt);
              if ((int)lane >= dlt) hs = max(hs, t);
            }
            S.res_lastw[tid] = hs;
          }
          __syncthreads();
        }
        P4LAP(0);
        // (b) every byte of the window looks up the match that covers it — the nearest start in its own 32 positions,
        // else the last one of the words before (a match that began in the window before is in res_strad) — and takes
        // that match's distance, or 0 for a literal byte.  The lane keeps its four entries in registers.
        uint32_t d[RES_W / PAR_THREADS];
        if (!scanmode) {
          // (the records of the windows before are gone — their bytes have been copied over them: the one match that can
          // reach in from there, the last one to start before this window, is carried in res_strad)
          const uint32_t sEnd = S.res_strad[0], sD = S.res_strad[1];
          const uint32_t* out32 = reinterpret_cast<const uint32_t*>(S.out);
          // (only this window's words are looked at: the words below it are no match starts any more in T2, which keeps its
          // marker flags there, and what reaches in from below is the carried match anyway)
          const uint32_t wlo = ws >> 5;
          uint32_t w0[RES_W / PAR_THREADS], w1[RES_W / PAR_THREADS];  // the byte's own word of the bitmap and the one before: all reads in flight
#pragma unroll
          for (uint32_t k = 0; k < RES_W / PAR_THREADS; k++) {
            const uint32_t wi = (ws + tid + k * PAR_THREADS) >> 5;
            w0[k] = S.bitmap[wi];
            w1[k] = S.bitmap[wi > wlo ? wi - 1u : wlo];
          }
#pragma unroll
          for (uint32_t k = 0; k < RES_W / PAR_THREADS; k++) {
            const uint32_t bpos = tid + k * PAR_THREADS, P = ws + bpos, wi = P >> 5;
            uint32_t w = w0[k] & (0xFFFFFFFFu >> (31u - (P & 31u)));  // starts at or before this byte, in its word
            uint32_t bk = 0;
            if (!w && wi > wlo) {  // (text: a start every few bytes — the own word or the one before)
              w = w1[k];
              bk = 1;
            }
            // (longer matches have their start up to nine words back: four words a read)
            for (uint32_t stg = 0; stg < 2u && !w && wi >= wlo + 2u + 4u * stg; stg++) {
              const uint32_t hiw = wi - 2u - 4u * stg, a = hiw >= wlo + 3u ? hiw - 3u : wlo;  // words [a, a + 3], those up to hiw count
              uint32_t q[4];
              __builtin_memcpy(q, &S.bitmap[a], 16);
#pragma unroll
              for (int t = 3; t >= 0; t--) {
                if (!w && a + (uint32_t)t <= hiw && q[t]) {
                  w = q[t];
                  bk = wi - (a + (uint32_t)t);
                }
              }
            }
            uint32_t dv = 0, e = 0, dd = 0;
            if (w) {
              const uint32_t s = 32u * (wi - bk) + 31u - (uint32_t)__clz(w);  // (inside this window: its record is whole)
              const uint32_t lo = out32[s >> 2], hi = out32[(s >> 2) + 1u];
              const uint32_t rec = __builtin_amdgcn_alignbyte(hi, lo, s & 3u);  // distance - 1 (16 bits), length - 3 (8 bits)
              const uint32_t D = (rec & 0xffffu) + 1u, L = ((rec >> 16) & 0xffu) + 3u;
              dv = P < s + L ? D : 0u;
              e = s + L;
              dd = D;
            } else {
              dv = P < sEnd ? sD : 0u;  // no start in this window in front of the byte: the match carried in, if it reaches
            }
            // the match that runs past this window's end, for the next window: what the window's last byte has found
            if (bpos == RES_W - 1u) {
              S.res_strad[2] = e > ws + RES_W ? e : 0u;
              S.res_strad[3] = dd;
            }
            dv = bpos < wlen ? dv : 0u;
            d[k] = dv;
            dist[bpos] = (uint16_t)dv;
          }
        } else {
          const uint32_t sEnd = S.res_strad[0], sD = S.res_strad[1];  // absolute end and distance of the match that straddles in
          const uint32_t* out32 = reinterpret_cast<const uint32_t*>(S.out);
#pragma unroll
          for (uint32_t k = 0; k < RES_W / PAR_THREADS; k++) {
            const uint32_t bpos = tid + k * PAR_THREADS, wi = bpos >> 5;
            const uint32_t w = S.bitmap[(ws >> 5) + wi] & (0xFFFFFFFFu >> (31u - (bpos & 31u)));  // starts at or before this byte
            // (the first 64 words' running maxima were made by wave 0, the others by wave 1: take both)
            uint32_t before = wi ? S.res_lastw[wi - 1u] : 0u;
            if (wi > 64u) before = max(before, S.res_lastw[63]);
            const uint32_t sp1 = w ? 32u * wi + 32u - (uint32_t)__clz(w) : before;  // start + 1, 0: no start in this window
            const uint32_t s = ws + (sp1 ? sp1 - 1u : 0u);
            const uint32_t lo = out32[s >> 2], hi = out32[(s >> 2) + 1u];
            const uint32_t rec = __builtin_amdgcn_alignbyte(hi, lo, s & 3u);  // distance - 1 (16 bits), length - 3 (8 bits)
            const uint32_t D = (rec & 0xffffu) + 1u, L = ((rec >> 16) & 0xffu) + 3u;
            const uint32_t P = ws + bpos;
            uint32_t dv = (sp1 && P < s + L) ? D : 0u;
            dv = (!sp1 && P < sEnd) ? sD : dv;
            dv = bpos < wlen ? dv : 0u;
            d[k] = dv;
            dist[bpos] = (uint16_t)dv;
          }
          // the match that runs past this window's end, for the next window
          if (tid == 0) {
            const uint32_t sp1 = max(S.res_lastw[63], S.res_lastw[RES_W / 32u - 1u]);
            uint32_t e = 0, dd = 0;
            if (sp1) {
              const uint32_t s = ws + sp1 - 1u;
              dd = ((uint32_t)S.out[s] | ((uint32_t)S.out[s + 1] << 8)) + 1u;
              e = s + (uint32_t)S.out[s + 2] + 3u;
            } else if (sEnd > ws + RES_W) {  // (a straddler longer than a window cannot be: 258 < RES_W)
              e = sEnd;
              dd = sD;
            }
            S.res_strad[2] = e > ws + RES_W ? e : 0u;
            S.res_strad[3] = dd;
          }
        }
        __syncthreads();
        P4LAP(1);
        // (c) pointer jumping until nothing moves.  A lane owns the entries of its bytes (only it writes them) and
        // keeps them in registers; an entry is live while its ancestor is an unresolved byte of this window.  Updates
        // are in place: whatever a lane reads from another byte's entry is the distance to one of that byte's
        // ancestors, at any moment.  RES_REPS steps per barrier (progress travels through LDS without one; the barrier
        // is there to find out that nothing moves any more); a wave whose bytes are settled skips its steps.
        bool live[RES_W / PAR_THREADS];
#pragma unroll
        for (uint32_t k = 0; k < RES_W / PAR_THREADS; k++) live[k] = d[k] != 0u && d[k] <= tid + k * PAR_THREADS;  // (d > b: the ancestor lies below the window: final)
        for (uint32_t rnd = 0;; rnd++) {
          bool moved = false;
#pragma unroll
          for (uint32_t rep = 0; rep < RES_REPS; rep++) {
            bool anylive = false;
#pragma unroll
            for (uint32_t k = 0; k < RES_W / PAR_THREADS; k++) anylive = anylive || live[k];
            if (!__ballot(anylive)) break;  // this wave's bytes are settled
            uint32_t da[RES_W / PAR_THREADS];
#pragma unroll
            for (uint32_t k = 0; k < RES_W / PAR_THREADS; k++) da[k] = dist[live[k] ? tid + k * PAR_THREADS - d[k] : 0u];  // one LDS round trip for all
#pragma unroll
            for (uint32_t k = 0; k < RES_W / PAR_THREADS; k++) {
              const bool mv = live[k] && da[k] != 0u;
              d[k] += mv ? da[k] : 0u;
              if (mv) dist[tid + k * PAR_THREADS] = (uint16_t)d[k];
              moved = moved || mv;
              live[k] = mv && d[k] <= tid + k * PAR_THREADS;  // an ancestor that is final stays final
            }
          }
          // "is any entry still on its way" (an entry whose ancestor is final is done: no round just to see that nothing
          // moves any more): three flags in rotation, so that clearing one never meets a wave that is a step ahead
          bool unsettled = false;
#pragma unroll
          for (uint32_t k = 0; k < RES_W / PAR_THREADS; k++) unsettled = unsettled || live[k];
          (void)moved;
          if (__ballot(unsettled) && lane == 0) S.res_flag[rnd % 3u] = 1u;
          __syncthreads();
          const bool any = S.res_flag[rnd % 3u] != 0u;
          if (tid == 0) S.res_flag[(rnd + 2u) % 3u] = 0u;
          nrounds++;
          if (!any) break;
        }
        P4LAP(2);
        // (d) every match byte of the window takes its value from its final ancestor
        if (!FOREIGN) {
#pragma unroll
          for (uint32_t k = 0; k < RES_W / PAR_THREADS; k++) {
            const uint32_t b = tid + k * PAR_THREADS;
            if (d[k] != 0u) S.out[ws + b] = S.out[ws + b - d[k]];
          }
        } else {
          // T2: the ancestor may lie in front of the block — the byte is then a marker, 256 + its index in the 32 KiB
          // window before the block — or be such a marker byte of a window resolved earlier: its symbol is read back
          // from the store (this workgroup wrote it; the bitmap words of resolved windows hold "is a marker").  All
          // symbols of the window go to the symbol store, two bytes each.
          uint16_t* sym16 = reinterpret_cast<uint16_t*>(symp);
          uint32_t sy[RES_W / PAR_THREADS];
          {
            // (a marker's symbol comes back from the store: the lane's four reads are asked for together, clamped and
            // unconditional — one trip to the L2 per window instead of up to four in a row: 133k of a block's 610k cycles)
            uint32_t wd[RES_W / PAR_THREADS], vb[RES_W / PAR_THREADS], gis[RES_W / PAR_THREADS];
            int32_t abs_[RES_W / PAR_THREADS];
            bool fromstore[RES_W / PAR_THREADS];
#pragma unroll
            for (uint32_t k = 0; k < RES_W / PAR_THREADS; k++) {
              const uint32_t b = tid + k * PAR_THREADS, P = ws + b;  // (positions inside the chunk)
              const int32_t a = (int32_t)P - (int32_t)d[k];           // the ancestor, chunk-relative
              const int32_t ab = a + (int32_t)clo;                    // ... block-relative
              const uint32_t ai = a < 0 ? 0u : (uint32_t)a;
              const bool hist = ab < 0;                               // in front of the block: a marker
              const bool prior = !hist && a < 0;                      // in a chunk resolved before: its symbol is in the store
              const bool mk = a >= 0 && ai < ws && ((S.bitmap[ai >> 5] >> (ai & 31u)) & 1u);
              vb[k] = S.out[ai];
              fromstore[k] = mk || prior;
              gis[k] = fromstore[k] ? (uint32_t)ab : 0u;
              abs_[k] = ab;
              wd[k] = __hip_atomic_load(&symp[gis[k] >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (uint32_t k = 0; k < RES_W / PAR_THREADS; k++) {
              uint32_t v = fromstore[k] ? ((gis[k] & 1u) ? wd[k] >> 16 : wd[k] & 0xffffu) : vb[k];
              v = abs_[k] < 0 ? (uint32_t)(256 + (int32_t)ZES_WINDOW + abs_[k]) : v;
              sy[k] = v;
            }
          }
#pragma unroll
          for (uint32_t k = 0; k < RES_W / PAR_THREADS; k++) {
            const uint32_t b = tid + k * PAR_THREADS;
            if (b < wlen) {
              S.out[ws + b] = (uint8_t)sy[k];
              sym16[clo + ws + b] = (uint16_t)sy[k];
            }
            // marker flags over this window's (dead) match bits: lanes tid .. tid + 63 of a wave are 64 consecutive bytes
            const uint64_t mm = __ballot(b < wlen && sy[k] >= 256u);
            if (lane == 0) {
              S.bitmap[((ws + b) >> 5)] = (uint32_t)mm;
              S.bitmap[((ws + b) >> 5) + 1u] = (uint32_t)(mm >> 32);
            }
          }
          // (read back later by this workgroup only, past a barrier, with loads that go to the L2: same-CU accesses stay in
          // order, no device-wide fence — which would write the whole L2 back: measured 4.7x slower)
        }
        P4LAP(3);
        if (tid == 0) {
          S.res_strad[0] = S.res_strad[2];
          S.res_strad[1] = S.res_strad[3];
        }
        __syncthreads();  // (also: the copies above have read dist[] and the image before the next window rewrites them)
      }
      if (dbg && tid == 0) {
        unsigned long long* row = dbg + (size_t)blockIdx.x * ZES_PAR_DBG_ROW;
        for (int k = 0; k < 4; k++) row[24 + k] = tacc[k];
        row[28] = nrounds;
      }
#undef P4LAP
    }
    __syncthreads();
    if (FOREIGN) {
      clo += PAR_CHUNK;
      if (clo < total) goto next_chunk;
    }

    STAMP(6);
    if (!FOREIGN) {
      // ---- P5: flush ----
      uint8_t* dst = it.dst;
      const uint32_t nstore = (uint32_t)min((uint64_t)total, it.room);
      const uint32_t full = nstore >> 4;
      for (uint32_t i = tid; i < full; i += PAR_THREADS)
        reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(S.out)[i];
      for (uint32_t i = (full << 4) + tid; i < nstore; i += PAR_THREADS) dst[i] = S.out[i];
      if (tid == 0) {
        ZesCandRes r;
        r.end_bit = end_bit;
        r.out_len = total;
        r.flags = 1u | (S.bfinal ? 2u : 0u);
        *it.cres = r;
        if (it.cres_host) {
          *it.cres_host = r;
          *it.start_host = it.start;
        }
      }
    } else {
      // ---- T2: the map of the last 32 Ki symbols (in front of a shorter block: the markers themselves), the result ----
      // the work item that starts where this block ends (uniform binary search); none, and not the final block: the
      // next block is not on the list (a stored or fixed block, a start the search missed) — the wave decoder's case
      uint32_t next = 0;
      const bool fin = S.bfinal != 0u;
      if (!fin) {
        const uint32_t want = end_bit - 16u;
        uint32_t lo = 0, hi = it.ncand;
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (it.cand[mid] < want) lo = mid + 1; else hi = mid;
        }
        if (lo < it.ncand && it.cand[lo] == want) next = lo + 1u;
        if (next == 0u || end_bit >= limit) {
          // The block is decoded and its symbols are in the store; what is missing is the block behind it (a stored or
          // fixed one — a zlib stream's short last block — or a start the search missed).  With the symbols in the
          // item's own share the wave decoder takes the item over from here: flags 8 = "decoded up to end_bit, out_len
          // symbols stored" (k_inf_seg_scan preloads its ring from the store and appends); a lone wave that decodes this
          // block again, token by token, needs ~7 ms per 100 KiB of output.
          if (next == 0u && end_bit < limit && symp == it.sym) {
            if (tid == 0) {
              ZesSegRes r;
              r.end_bit = end_bit;
              r.out_len = total;
              r.flags = 8u;
              r.next = 0;
              *it.sres = r;
              it.fail_list[1u + atomicAdd(&it.fail_list[0], 1u)] = it.w;
            }
            return;
          }
          PAR_DECLINE(end_bit, total);
          return;
        }
      }
      for (uint32_t i2 = tid; i2 < ZES_WINDOW / 2; i2 += PAR_THREADS) {
        uint32_t v[2];
#pragma unroll
        for (uint32_t h = 0; h < 2; h++) {
          const int32_t pos = (int32_t)total - (int32_t)(ZES_WINDOW - (2u * i2 + h));
          if (pos < 0) {
            v[h] = (uint32_t)(256 + (int32_t)ZES_WINDOW + pos);
          } else {
            const uint32_t wd = __hip_atomic_load(&symp[(uint32_t)pos >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[h] = (pos & 1) ? wd >> 16 : wd & 0xffffu;
          }
        }
        it.map[i2] = v[0] | (v[1] << 16);
      }
      if (tid == 0) {
        ZesSegRes r;
        r.end_bit = end_bit;
        r.out_len = total;
        r.flags = 1u | (fin ? 2u : 0u);
        r.next = next;
        *it.sres = r;
      }
    }
    STAMP(7);
  }
}

#undef PAR_DECLINE
#undef STAMP

template <bool TWO>
__device__ __forceinline__ static void block_par_t1(ParSmem& S, const uint8_t* __restrict__ d_in, uint8_t* __restrict__ d_out,
                                                    const ZesInfBuf* __restrict__ bufs, uint32_t nbuf, const uint32_t* __restrict__ cnt,
                                                    const uint32_t* __restrict__ cand_all, const uint32_t* __restrict__ map_all,
                                                    ZesCandRes* __restrict__ cres_all, unsigned long long* __restrict__ dbg,
                                                    const uint32_t* __restrict__ redo, const uint32_t* __restrict__ raw_all,
                                                    uint32_t* __restrict__ sorted_all, const ZesParMirror& mir) {
  // (one-buffer calls: the search's counters go to the host with this kernel — they are final — so that the host can check
  // the chain of blocks itself after one synchronisation, without a chain kernel behind this one)
  if (mir.counters_host && blockIdx.x == 0)
    for (uint32_t i = threadIdx.x; i < mir.counter_words; i += PAR_THREADS) mir.counters_host[i] = mir.counters[i];
  // buffer of this work item: the last entry whose first work item is <= blockIdx.x
  uint32_t bi = 0;
  {
    uint32_t lo = 0, hi = nbuf;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (bufs[mid].work_first <= blockIdx.x) lo = mid; else hi = mid;
    }
    bi = lo;
  }
  // work item inside the buffer = output slot (a one-buffer launch may name the slots to decode again: redo[])
  uint32_t w = redo ? redo[blockIdx.x] : blockIdx.x - bufs[bi].work_first;
  const uint32_t ncand = min(cnt[bi], bufs[bi].cand_cap);
  uint32_t nwork = bufs[bi + 1].work_first - bufs[bi].work_first;
  if (bufs[nbuf].work_first == ZES_WORK_AUTO) {  // one buffer, launched before the host saw the candidate count
    nwork = min(ncand, bufs[nbuf].cand_cap);
    if (blockIdx.x >= nwork) return;
  }
  const uint32_t* cand = cand_all + bufs[bi].cand_base;
  const uint32_t* map = map_all ? map_all + bufs[bi].cand_base : nullptr;
  // First launch over a group (raw_all): the candidate list is still in the order the verify kernels found it.  Work
  // item x takes list entry x, ranks it — its rank is its block's number, hence its output slot — and looks up the two
  // candidates behind it (the end estimates): a pass over a list of a few hundred positions per workgroup instead of a
  // sorting kernel of one workgroup in front of this one (17 us + a launch boundary per call).  The sorted list, which
  // the chain check and the repair paths read, is written on the way.
  uint32_t w_rank = 0, raw_next = 0xFFFFFFFFu, raw_next2 = 0xFFFFFFFFu, raw_min = 0, raw_mine = 0;
  const bool ranked = raw_all && !map_all && !redo;
  if (ranked) {
    const uint32_t* raw = raw_all + bufs[bi].cand_base;
    const uint32_t x = blockIdx.x - bufs[bi].work_first;
    if (x >= ncand) return;
    const uint32_t u = raw[x];
    if (threadIdx.x == 0) {
      S.wave_sum[0] = 0;
      S.wave_sum[1] = 0xFFFFFFFFu;
      S.wave_sum[2] = 0xFFFFFFFFu;
      S.wave_sum[3] = 0xFFFFFFFFu;
    }
    __syncthreads();
    uint32_t less = 0, nx = 0xFFFFFFFFu, mn = 0xFFFFFFFFu;
    for (uint32_t j = threadIdx.x; j < ncand; j += PAR_THREADS) {
      const uint32_t v = raw[j];
      less += (v < u || (v == u && j < x)) ? 1u : 0u;
      nx = (v > u && v < nx) ? v : nx;
      mn = min(mn, v);
    }
    if (less) atomicAdd(&S.wave_sum[0], less);
    if (nx != 0xFFFFFFFFu) atomicMin(&S.wave_sum[1], nx);
    atomicMin(&S.wave_sum[3], mn);
    __syncthreads();
    raw_next = S.wave_sum[1];
    uint32_t nx2 = 0xFFFFFFFFu;
    for (uint32_t j = threadIdx.x; j < ncand; j += PAR_THREADS) {
      const uint32_t v = raw[j];
      nx2 = (v > raw_next && v < nx2) ? v : nx2;
    }
    if (raw_next != 0xFFFFFFFFu && nx2 != 0xFFFFFFFFu) atomicMin(&S.wave_sum[2], nx2);
    __syncthreads();
    w_rank = S.wave_sum[0];
    raw_next2 = S.wave_sum[2];
    raw_min = S.wave_sum[3];
    raw_mine = u;
    if (threadIdx.x == 0) {
      sorted_all[bufs[bi].cand_base + w_rank] = u;
      // the host's mirror: the start of EVERY candidate and a cleared result, before anything below can leave (the list not
      // beginning at the stream's first block: nobody decodes) — what the host reads is this launch's or nothing
      if (mir.cres_host) {
        ZesCandRes r0;
        r0.end_bit = 0;
        r0.out_len = 0;
        r0.flags = 0;
        mir.cres_host[w_rank] = r0;
        mir.start_host[w_rank] = u + 16u;
      }
    }
    __syncthreads();  // (the words are used again further down)
  }
  // the first block of a reference-made stream starts at bit 16 and passes the candidate rules: if the sorted
  // list does not begin there, k_inf_chain rejects the buffer whatever is decoded here (another encoder's stream)
  if (!map_all && !redo && (((ranked ? raw_min : cand[0]) != bufs[bi].start_rel && !(bufs[bi].range_flags & ZES_START_ANY)) || cnt[bi] > bufs[bi].cand_cap)) return;  // (or more candidates than the output has blocks)
  const uint64_t c = bufs[bi].c;
  if (ranked) w = w_rank;
  const uint32_t ci = map ? map[w] : w;
  ParItem it;
  it.g32 = reinterpret_cast<const uint32_t*>(d_in + bufs[bi].in_off);
  it.lastdw = (uint32_t)((c - 1) >> 2);
  it.limit = (uint32_t)(c * 8);
  it.start = (ranked ? raw_mine : cand[ci]) + 16u;
  it.de_est = it.limit;
  it.de_est2 = 0;
  if (ranked) {
    if (raw_next != 0xFFFFFFFFu) {
      it.de_est = raw_next + 16u;
      it.de_est2 = raw_next2 != 0xFFFFFFFFu ? raw_next2 + 16u : it.limit;
    }
  } else if (map) {
    if (w + 1 < nwork) it.de_est = cand[map[w + 1]] + 16u;
  } else if (ci + 1u < ncand) {
    it.de_est = cand[ci + 1u] + 16u;
    it.de_est2 = ci + 2u < ncand ? cand[ci + 2u] + 16u : it.limit;
  }
  it.cres = cres_all + bufs[bi].cand_base + w;
  it.cres_host = mir.cres_host ? mir.cres_host + w : nullptr;  // (one buffer: its candidates start at 0)
  it.start_host = mir.cres_host ? mir.start_host + w : nullptr;
  const uint64_t slot_off = (uint64_t)w * ZES_BLK;
  it.dst = d_out + bufs[bi].out_off + slot_off;
  it.room = bufs[bi].cap > slot_off ? bufs[bi].cap - slot_off : 0;
  it.sres = nullptr;
  it.sym = nullptr;
  it.sym_cap = 0;
  it.sym_all = nullptr;
  it.bump = nullptr;
  it.bump_base = it.bump_cap = 0;
  it.symoff = nullptr;
  it.map = nullptr;
  it.cand = nullptr;
  it.ncand = 0;
  it.fail_list = nullptr;
  it.w = w;
  par_body<false, TWO>(S, it, dbg);
}

__global__ __launch_bounds__(PAR_THREADS) void k_inf_block_par(const uint8_t* __restrict__ d_in, uint8_t* __restrict__ d_out,
                                                               const ZesInfBuf* __restrict__ bufs, uint32_t nbuf,
                                                               const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ cand_all,
                                                               const uint32_t* __restrict__ map_all, ZesCandRes* __restrict__ cres_all,
                                                               unsigned long long* __restrict__ dbg, const uint32_t* __restrict__ redo,
                                                               const uint32_t* __restrict__ raw_all, uint32_t* __restrict__ sorted_all, ZesParMirror mir) {
  __shared__ __align__(16) ParSmem S;
  block_par_t1<false>(S, d_in, d_out, bufs, nbuf, cnt, cand_all, map_all, cres_all, dbg, redo, raw_all, sorted_all, mir);
}
// the same for compressible data (the launch's streams are shorter than 0.7 of their outputs' capacity)
__global__ __launch_bounds__(PAR_THREADS) void k_inf_block_par2(const uint8_t* __restrict__ d_in, uint8_t* __restrict__ d_out,
                                                                const ZesInfBuf* __restrict__ bufs, uint32_t nbuf,
                                                                const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ cand_all,
                                                                const uint32_t* __restrict__ map_all, ZesCandRes* __restrict__ cres_all,
                                                                unsigned long long* __restrict__ dbg, const uint32_t* __restrict__ redo,
                                                                const uint32_t* __restrict__ raw_all, uint32_t* __restrict__ sorted_all, ZesParMirror mir) {
  __shared__ __align__(16) ParSmem S;
  block_par_t1<true>(S, d_in, d_out, bufs, nbuf, cnt, cand_all, map_all, cres_all, dbg, redo, raw_all, sorted_all, mir);
}

// T2: one workgroup per block of another encoder's stream (work items as in k_inf_seg_scan: a buffer's item 0 starts at
// bit 16, its item w at candidate w - 1).
__global__ __launch_bounds__(PAR_THREADS) void k_inf_seg_block_par(const uint8_t* __restrict__ d_in, const ZesSegJob* __restrict__ jobs,
                                                                   uint32_t njobs, const uint32_t* __restrict__ cand_all,
                                                                   ZesSegRes* __restrict__ sres_all, uint32_t* __restrict__ maps_all,
                                                                   uint32_t* __restrict__ sym16_all, uint32_t sym_ratio,
                                                                   uint32_t* __restrict__ fail_list, unsigned long long* __restrict__ bump,
                                                                   uint64_t bump_base, uint64_t bump_cap, uint64_t* __restrict__ symoff,
                                                                   unsigned long long* __restrict__ dbg) {
  __shared__ __align__(16) ParSmem S;
  // buffer of this work item: the last one whose first work item is <= blockIdx.x
  uint32_t bi = 0;
  {
    uint32_t lo = 0, hi = njobs;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (jobs[mid].work_first <= blockIdx.x) lo = mid; else hi = mid;
    }
    bi = lo;
  }
  const ZesSegJob jb = jobs[bi];
  const uint32_t w = blockIdx.x - jb.work_first;
  const uint32_t ncand = jb.ncand;
  const uint32_t* cand = cand_all + jb.cand_base;
  ZesSegRes* sres = sres_all + jb.work_first;
  const uint64_t c = jb.c;
  uint64_t start = jb.start0;
  if (w > 0) {
    const uint32_t c0 = cand[w - 1];
    if (c0 + 16u == jb.start0) {  // the stream start is work item 0 already
      if (threadIdx.x == 0) {
        ZesSegRes r;
        r.end_bit = 0;
        r.out_len = 0;
        r.flags = 0;
        r.next = 0;
        sres[w] = r;
      }
      return;
    }
    start = (uint64_t)c0 + 16;
  }
  uint32_t nx = w;  // candidate that starts the next work item (candidate 0 at bit 16 duplicates work item 0)
  if (w == 0 && ncand > 0 && cand[0] + 16u == jb.start0) nx = 1;
  ParItem it;
  it.g32 = reinterpret_cast<const uint32_t*>(d_in + jb.in_off);
  it.lastdw = (uint32_t)((c - 1) >> 2);
  it.limit = (uint32_t)(c * 8);
  it.start = (uint32_t)start;
  it.de_est = nx < ncand ? cand[nx] + 16u : it.limit;
  it.de_est2 = nx < ncand ? (nx + 1u < ncand ? cand[nx + 1u] + 16u : it.limit) : 0u;
  it.cres = nullptr;
  it.cres_host = nullptr;
  it.start_host = nullptr;
  it.dst = nullptr;
  it.room = 0;
  it.sres = sres + w;
  // symbol store: the work item that starts at compressed byte b owns symbols [b * ratio, b' * ratio), b' the start of
  // the next work item (the end of the stream for the last one)
  const uint64_t b0 = (start - 16) / 8, b1 = nx < ncand ? (uint64_t)cand[nx] / 8 : c;
  it.sym = sym16_all + jb.sym_base + b0 * sym_ratio / 2;
  it.sym_cap = (b1 - b0) * sym_ratio;
  it.sym_all = sym16_all;
  it.bump = bump;
  it.bump_base = bump_base;
  it.bump_cap = bump_cap;
  it.symoff = symoff + blockIdx.x;
  it.map = maps_all + (size_t)jb.work_first * (ZES_WINDOW / 2) + (size_t)w * (ZES_WINDOW / 2);
  it.cand = cand;
  it.ncand = ncand;
  it.fail_list = fail_list;
  it.w = blockIdx.x;  // (the list numbers items over the whole group)
  par_body<true, true>(S, it, dbg);
}
