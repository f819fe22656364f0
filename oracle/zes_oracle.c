/*
 * zes_oracle.c — CPU restatement of zprodev/zlib.es v0.6.0 (deflate/inflate hot path).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped path (zlib.es_amd/, the C-ABI library,
 * the N-API addon) may include, link or call this file.  Its users are tests/, the checker in
 * __graft_entry__.smoke() and the `cpu_baseline` leg of bench.py.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 *   (a) every vector the reference's own suite holds (test/index.js:7-10,16-42,57-108),
 *   (b) fixtures produced by running the reference bundle under Node in the build container
 *       (tests/golden/make_golden.js → tests/golden/), incl. stage-level LZ77 token lists and
 *       package-merge code lengths, error cases and truncated/corrupted streams,
 *   (c) the live reference when /root/reference and node are present (skipped otherwise).
 *
 * One deliberate difference, on inputs the reference itself cannot finish: readRange() refills with zero bits
 * past the end of the data without setting isEnd (src/utils/BitReadStream.ts:33-42), so a truncated stream whose
 * zero bits decode as (length, distance) tokens with extra bits is decoded for ever — the reference ends with a
 * RangeError when memory runs out.  Past the end every bit is zero, so the token loop is a cycle over the
 * handful of reader states: once the reader is 1 KiB past the end (or, as a second fence, the output has passed
 * DEFLATE's 1032:1 limit + 64 KiB) the oracle — and the product path's exact decoder, the same way — stops
 * with 'Lack of data length'.
 *
 * Single-threaded plain C.  Each function names the reference lines it follows
 * (paths relative to /root/reference).  The structure (arrays, radix-sorted index instead of
 * JS objects) is this repo's own; only behaviour is restated.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ZOR_OK 0
#define ZOR_E_NOT_DEFLATE (-1)
#define ZOR_E_BTYPE3 (-2)
#define ZOR_E_CORRUPT (-3)
#define ZOR_E_INSUFFICIENT (-4)
#define ZOR_E_LACK (-5)
#define ZOR_E_NOSPACE (-16)
#define ZOR_E_ARG (-18)

#define BLOCK_MAX 131072u /* src/const.ts:7 */

/* src/const.ts:9-35 (RFC1951 tables) */
static const uint8_t LEN_XBITS[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                      2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t LEN_BASE[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                      31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint16_t DIST_BASE[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,
                                       33,  49,  65,  97,  129, 193,  257,  385,  513,  769,
                                       1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_XBITS[30] = {0, 0, 0, 0, 1, 1, 2, 2,  3,  3,  4,  4,  5,  5,  6,
                                       6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t CODELEN_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

/* ------------------------------------------------------------------------------------------
 * Adler-32 — src/adler32.ts:1-10.  The reference returns (s2<<16)+s1 as a JS int32 and
 * extracts bytes with >>> (src/zlib.ts:37-40), i.e. the standard unsigned value.
 * ---------------------------------------------------------------------------------------- */
int zor_adler32(const uint8_t* in, uint64_t n, uint32_t* out) {
  uint32_t s1 = 1, s2 = 0;
  for (uint64_t i = 0; i < n; i++) {
    s1 = (s1 + in[i]) % 65521u;
    s2 = (s1 + s2) % 65521u;
  }
  *out = (s2 << 16) | s1;
  return ZOR_OK;
}

/* ------------------------------------------------------------------------------------------
 * LZ77 — src/lz77.ts:11-119.
 * Token encoding (shared with include/zes.h): literal byte, or 0x80000000|(len-3)<<16|(dist-1).
 * ---------------------------------------------------------------------------------------- */
static inline int in_at(const uint8_t* in, uint64_t n, uint64_t i) {
  /* JS typed-array read: out-of-range index yields `undefined`; undefined !== undefined is
   * false, so two out-of-range reads compare equal (src/lz77.ts:73,81).  -1 models that. */
  return i < n ? (int)in[i] : -1;
}

/* Index of the block: positions start..start+len-3 grouped by exact 3-byte key, each group in
 * ascending position order (src/lz77.ts:11-22: one array per key, pushed in scan order).
 * Built as a stable LSD radix sort instead of a JS object of arrays. */
typedef struct {
  uint32_t* sorted; /* sorted[r] = position - start */
  uint32_t* rank;   /* rank[pos-start] = r */
  uint32_t count;
} zor_index;

static inline uint32_t key3(const uint8_t* in, uint64_t i) {
  return ((uint32_t)in[i] << 16) | ((uint32_t)in[i + 1] << 8) | in[i + 2];
}

static int build_index(const uint8_t* in, uint64_t start, uint32_t len, zor_index* ix) {
  ix->count = len >= 3 ? len - 2 : 0;
  uint32_t m = ix->count ? ix->count : 1;
  ix->sorted = (uint32_t*)malloc(sizeof(uint32_t) * m);
  ix->rank = (uint32_t*)malloc(sizeof(uint32_t) * m);
  uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * m);
  if (!ix->sorted || !ix->rank || !tmp) return ZOR_E_ARG;
  for (uint32_t i = 0; i < ix->count; i++) ix->sorted[i] = i;
  uint32_t* a = ix->sorted;
  uint32_t* b = tmp;
  for (int pass = 0; pass < 3; pass++) {
    uint32_t hist[257];
    memset(hist, 0, sizeof hist);
    int off = 2 - pass; /* least significant key byte first */
    for (uint32_t i = 0; i < ix->count; i++) hist[in[start + a[i] + off] + 1]++;
    for (int v = 0; v < 256; v++) hist[v + 1] += hist[v];
    for (uint32_t i = 0; i < ix->count; i++) b[hist[in[start + a[i] + off]]++] = a[i];
    uint32_t* t = a;
    a = b;
    b = t;
  }
  /* 3 passes: result is in tmp (a == tmp) */
  memcpy(ix->sorted, a, sizeof(uint32_t) * ix->count);
  for (uint32_t r = 0; r < ix->count; r++) ix->rank[ix->sorted[r]] = r;
  free(tmp);
  return ZOR_OK;
}

static void free_index(zor_index* ix) {
  free(ix->sorted);
  free(ix->rank);
}

/* generateLZ77Codes — src/lz77.ts:24-119.  tokens must hold len entries (+2 slack). */
int zor_lz77_block(const uint8_t* in, uint64_t n, uint64_t start, uint32_t len, uint32_t* tokens,
                   uint32_t* ntokens) {
  if (len < 2 || start + len > n) return ZOR_E_CORRUPT; /* undefined tokens → throw */
  zor_index ix;
  int rc = build_index(in, start, len, &ix);
  if (rc) return rc;
  uint32_t nt = 0;
  int64_t now = (int64_t)start;
  const int64_t end_index = (int64_t)start + (int64_t)len - 3; /* :26 */
  while (now <= end_index) {                                   /* :39 */
    const uint32_t r = ix.rank[now - start];
    const uint32_t k = key3(in, (uint64_t)now);
    const int64_t slide_base = now > 0x8000 ? now - 0x8000 : 0; /* :49 */
    int best = 0;
    int64_t best_idx = 0;
    int check = 0;
    /* candidates: same key, position < now, >= slide_base, most recent first (:65).  The
     * start/endIndexMap cursors (:53-62) only cache where that range begins and ends. */
    for (int64_t rr = (int64_t)r - 1; rr >= 0; rr--) {
      const int64_t q = (int64_t)start + ix.sorted[rr];
      if (key3(in, (uint64_t)q) != k) break;
      if (q < slide_base) break;
      if (check >= 128 || (best >= 8 && check >= 16)) break; /* :66-69 */
      check++;                                               /* :70 */
      int skip = 0;
      for (int j = best - 1; j > 0; j--) { /* :72-76 */
        if (in_at(in, n, (uint64_t)(q + j)) != in_at(in, n, (uint64_t)(now + j))) {
          skip = 1;
          break;
        }
      }
      if (skip) continue;
      int rl = 258;                        /* :78 */
      for (int j = best; j <= 258; j++) { /* :80-85 */
        if (in_at(in, n, (uint64_t)(q + j)) != in_at(in, n, (uint64_t)(now + j))) {
          rl = j;
          break;
        }
      }
      if (best < rl) { /* :86-92 */
        best = rl;
        best_idx = q;
        if (rl >= 258) break;
      }
    }
    if (best >= 3 && now + best <= end_index) { /* :95 */
      uint32_t dist = (uint32_t)(now - best_idx);
      tokens[nt++] = 0x80000000u | ((uint32_t)(best - 3) << 16) | (dist - 1);
      now += best;
    } else {
      tokens[nt++] = in[now];
      now++;
    }
  }
  tokens[nt++] = in[now];     /* :116 */
  tokens[nt++] = in[now + 1]; /* :117 */
  *ntokens = nt;
  free_index(&ix);
  return ZOR_OK;
}

static inline int len_code(int len) { /* src/lz77.ts:97-102 */
  int c = 0;
  for (int i = 0; i < 29; i++) {
    if (LEN_BASE[i] > len) break;
    c = i;
  }
  return c;
}
static inline int dist_code(int dist) { /* src/lz77.ts:103-108 */
  int c = 0;
  for (int i = 0; i < 30; i++) {
    if (DIST_BASE[i] > dist) break;
    c = i;
  }
  return c;
}

/* ------------------------------------------------------------------------------------------
 * Length-limited Huffman lengths by package-merge — src/huffman.ts:55-115, restated with
 * explicit symbol lists exactly as the reference keeps them (`simbles`).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  uint64_t count;
  uint32_t off; /* into the level's symbol pool */
  uint32_t nsym;
} zor_pack;

/* stable merge sort by count (V8 >= 7.0 Array.prototype.sort is stable: SURVEY A.4) */
static void stable_sort_packs(zor_pack* a, zor_pack* tmp, uint32_t n) {
  for (uint32_t w = 1; w < n; w *= 2) {
    for (uint32_t lo = 0; lo < n; lo += 2 * w) {
      uint32_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
      uint32_t i = lo, j = mid, k = lo;
      while (i < mid && j < hi) tmp[k++] = (a[j].count < a[i].count) ? a[j++] : a[i++];
      while (i < mid) tmp[k++] = a[i++];
      while (j < hi) tmp[k++] = a[j++];
    }
    memcpy(a, tmp, sizeof(zor_pack) * n);
  }
}

int zor_huff_lengths(const uint32_t* hist, uint32_t nsym, uint32_t maxlen, uint8_t* lens) {
  memset(lens, 0, nsym);
  uint32_t nkeys = 0;
  for (uint32_t s = 0; s < nsym; s++)
    if (hist[s]) nkeys++;
  if (nkeys == 0) return ZOR_OK; /* empty table (src/deflate.ts:90-97 then writes one 0) */
  if (nkeys == 1) {              /* :71-75 */
    for (uint32_t s = 0; s < nsym; s++)
      if (hist[s]) lens[s] = 1;
    return ZOR_OK;
  }
  /* per level: at most nkeys + nkeys items; symbols per level <= nkeys * level */
  const uint32_t max_items = 2 * nkeys + 2;
  const uint32_t pool_cap = nkeys * (maxlen + 1) + 16;
  zor_pack* prev = (zor_pack*)malloc(sizeof(zor_pack) * max_items);
  zor_pack* cur = (zor_pack*)malloc(sizeof(zor_pack) * max_items);
  zor_pack* tmp = (zor_pack*)malloc(sizeof(zor_pack) * max_items);
  uint16_t* pool_prev = (uint16_t*)malloc(sizeof(uint16_t) * pool_cap * 2);
  uint16_t* pool_cur = (uint16_t*)malloc(sizeof(uint16_t) * pool_cap * 2);
  uint32_t nprev = 0, ncur = 0;
  for (uint32_t level = 0; level < maxlen; level++) { /* :77 */
    ncur = 0;
    uint32_t used = 0;
    for (uint32_t s = 0; s < nsym; s++) { /* :79-85 leaves in ascending symbol order */
      if (!hist[s]) continue;
      cur[ncur].count = hist[s];
      cur[ncur].off = used;
      cur[ncur].nsym = 1;
      pool_cur[used++] = (uint16_t)s;
      ncur++;
    }
    for (uint32_t i = 0; i + 2 <= nprev; i += 2) { /* :87-94 */
      cur[ncur].count = prev[i].count + prev[i + 1].count;
      cur[ncur].off = used;
      cur[ncur].nsym = prev[i].nsym + prev[i + 1].nsym;
      memcpy(pool_cur + used, pool_prev + prev[i].off, sizeof(uint16_t) * prev[i].nsym);
      used += prev[i].nsym;
      memcpy(pool_cur + used, pool_prev + prev[i + 1].off, sizeof(uint16_t) * prev[i + 1].nsym);
      used += prev[i + 1].nsym;
      ncur++;
    }
    stable_sort_packs(cur, tmp, ncur); /* :95-99 */
    if (ncur % 2 != 0) ncur--;          /* :100-102 */
    zor_pack* tp = prev;
    prev = cur;
    cur = tp;
    uint16_t* tq = pool_prev;
    pool_prev = pool_cur;
    pool_cur = tq;
    nprev = ncur;
  }
  /* :106-115 — code length = number of occurrences over all items of the final list */
  for (uint32_t i = 0; i < nprev; i++)
    for (uint32_t k = 0; k < prev[i].nsym; k++) lens[pool_prev[prev[i].off + k]]++;
  free(prev);
  free(cur);
  free(tmp);
  free(pool_prev);
  free(pool_cur);
  return ZOR_OK;
}

/* canonical codes from lengths — src/huffman.ts:117-151 */
static void canon_codes(const uint8_t* lens, uint32_t nsym, uint16_t* codes) {
  int lmin = 99, lmax = 0;
  for (uint32_t s = 0; s < nsym; s++)
    if (lens[s]) {
      if (lens[s] < lmin) lmin = lens[s];
      if (lens[s] > lmax) lmax = lens[s];
    }
  uint32_t code = 0;
  for (int l = lmin; l <= lmax; l++) {
    for (uint32_t s = 0; s < nsym; s++)
      if (lens[s] == l) codes[s] = (uint16_t)code++;
    code <<= 1;
  }
}

/* ------------------------------------------------------------------------------------------
 * Bit writer — src/utils/BitWriteStream.ts:1-47
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  uint8_t* buf;
  uint64_t cap, idx;
  uint32_t now_bits, now_idx;
  int is_end, err;
} zor_bw;

static void bw_write(zor_bw* w, int bit) { /* :14-28 */
  if (w->is_end) {
    w->err = ZOR_E_LACK;
    return;
  }
  w->now_bits += (uint32_t)bit << w->now_idx;
  w->now_idx++;
  if (w->now_idx >= 8) {
    w->buf[w->idx++] = (uint8_t)w->now_bits;
    w->now_bits = 0;
    w->now_idx = 0;
    if (w->cap <= w->idx) w->is_end = 1;
  }
}
static void bw_range(zor_bw* w, uint32_t value, int length) { /* :29-37 LSB first */
  for (int i = 0; i < length; i++) bw_write(w, (value >> i) & 1);
}
static void bw_coded(zor_bw* w, uint32_t value, int length) { /* :38-46 MSB first */
  for (int i = length - 1; i >= 0; i--) bw_write(w, (value >> i) & 1);
}

/* deflateDynamicBlock — src/deflate.ts:56-227 */
static int deflate_dynamic_block(zor_bw* w, const uint8_t* in, uint64_t n, uint64_t start, uint32_t len,
                                 uint32_t* tokens) {
  uint32_t nt = 0;
  int rc = zor_lz77_block(in, n, start, len, tokens, &nt); /* :57 */
  if (rc) return rc;
  uint32_t lhist[286], dhist[30];
  memset(lhist, 0, sizeof lhist);
  memset(dhist, 0, sizeof dhist);
  lhist[256] = 1; /* :58 */
  int lmax = 256, dmax = 0;
  for (uint32_t i = 0; i < nt; i++) { /* :62-77 */
    uint32_t t = tokens[i];
    if (t & 0x80000000u) {
      int l = (int)((t >> 16) & 0xff) + 3, d = (int)(t & 0x7fff) + 1;
      int lc = len_code(l) + 257, dc = dist_code(d);
      lhist[lc]++;
      dhist[dc]++;
      if (lc > lmax) lmax = lc;
      if (dc > dmax) dmax = dc;
    } else {
      lhist[t]++;
    }
  }
  uint8_t llens[286], dlens[30];
  uint16_t lcodes[286], dcodes[30];
  zor_huff_lengths(lhist, 286, 15, llens); /* :78 */
  zor_huff_lengths(dhist, 30, 15, dlens);  /* :79 */
  canon_codes(llens, 286, lcodes);
  canon_codes(dlens, 30, dcodes);

  uint8_t codelens[320];
  int ncl = 0;
  for (int i = 0; i <= lmax; i++) codelens[ncl++] = llens[i]; /* :82-88 */
  const int HLIT = ncl;
  for (int i = 0; i <= dmax; i++) codelens[ncl++] = dlens[i]; /* :90-96 */
  const int HDIST = ncl - HLIT;

  uint8_t rl_codes[320];
  uint8_t rl_rep[320];
  int nrl = 0;
  for (int i = 0; i < ncl; i++) { /* :103-139 */
    int cl = codelens[i], rep = 1;
    while (i + 1 < ncl && cl == codelens[i + 1]) {
      rep++;
      i++;
      if (cl == 0) {
        if (138 <= rep) break;
      } else {
        if (6 <= rep) break;
      }
    }
    if (4 <= rep) {
      if (cl == 0) {
        rl_codes[nrl] = (11 <= rep) ? 18 : 17;
      } else {
        rl_codes[nrl] = (uint8_t)cl;
        rl_rep[nrl] = 1;
        nrl++;
        rep--;
        rl_codes[nrl] = 16;
      }
      rl_rep[nrl] = (uint8_t)rep;
      nrl++;
    } else {
      for (int j = 0; j < rep; j++) {
        rl_codes[nrl] = (uint8_t)cl;
        rl_rep[nrl] = 1;
        nrl++;
      }
    }
  }
  uint32_t chist[19];
  memset(chist, 0, sizeof chist);
  for (int i = 0; i < nrl; i++) chist[rl_codes[i]]++;
  uint8_t clens[19];
  uint16_t ccodes[19];
  zor_huff_lengths(chist, 19, 7, clens); /* :141 */
  canon_codes(clens, 19, ccodes);
  int HCLEN = 0;
  for (int i = 0; i < 19; i++) /* :143-148 */
    if (clens[CODELEN_ORDER[i]]) HCLEN = i + 1;

  bw_range(w, (uint32_t)(HLIT - 257), 5); /* :151 */
  bw_range(w, (uint32_t)(HDIST - 1), 5);  /* :153 */
  bw_range(w, (uint32_t)(HCLEN - 4), 4);  /* :155 */
  for (int i = 0; i < HCLEN; i++) bw_range(w, clens[CODELEN_ORDER[i]], 3); /* :158-165 */
  for (int i = 0; i < nrl; i++) {                                          /* :167-181 */
    int v = rl_codes[i];
    if (!clens[v]) return ZOR_E_CORRUPT;
    bw_coded(w, ccodes[v], clens[v]);
    if (v == 18) bw_range(w, (uint32_t)(rl_rep[i] - 11), 7);
    else if (v == 17) bw_range(w, (uint32_t)(rl_rep[i] - 3), 3);
    else if (v == 16) bw_range(w, (uint32_t)(rl_rep[i] - 3), 2);
  }
  for (uint32_t i = 0; i < nt; i++) { /* :183-220 */
    uint32_t t = tokens[i];
    if (t & 0x80000000u) {
      int l = (int)((t >> 16) & 0xff) + 3, d = (int)(t & 0x7fff) + 1;
      int lc = len_code(l), dc = dist_code(d);
      if (!llens[lc + 257]) return ZOR_E_CORRUPT;
      bw_coded(w, lcodes[lc + 257], llens[lc + 257]);
      if (LEN_XBITS[lc]) bw_range(w, (uint32_t)(l - LEN_BASE[lc]), LEN_XBITS[lc]);
      if (!dlens[dc]) return ZOR_E_CORRUPT;
      bw_coded(w, dcodes[dc], dlens[dc]);
      if (DIST_XBITS[dc]) bw_range(w, (uint32_t)(d - DIST_BASE[dc]), DIST_XBITS[dc]);
    } else {
      if (!llens[t]) return ZOR_E_CORRUPT;
      bw_coded(w, lcodes[t], llens[t]);
    }
  }
  if (!llens[256]) return ZOR_E_CORRUPT; /* :222-226 */
  bw_coded(w, lcodes[256], llens[256]);
  return w->err;
}

int zor_deflate_bound(uint64_t n, uint64_t* cap) { /* src/deflate.ts:16 + src/zlib.ts:42 */
  *cap = ((n < BLOCK_MAX / 2) ? BLOCK_MAX : n * 2) + 6;
  return ZOR_OK;
}

/* raw deflate — src/deflate.ts:14-39 */
int zor_deflate_raw(const uint8_t* in, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* out_len) {
  const uint64_t heap = (n < BLOCK_MAX / 2) ? BLOCK_MAX : n * 2; /* :16 */
  if (cap < heap) return ZOR_E_NOSPACE;
  memset(out, 0, heap);
  zor_bw w = {out, heap, 0, 0, 0, 0, 0};
  uint32_t* tokens = (uint32_t*)malloc(sizeof(uint32_t) * (BLOCK_MAX + 2));
  uint64_t processed = 0;
  int rc = ZOR_OK;
  while (1) { /* :20-34 */
    uint32_t target;
    if (processed + BLOCK_MAX >= n) {
      target = (uint32_t)(n - processed);
      bw_range(&w, 1, 1);
    } else {
      target = BLOCK_MAX;
      bw_range(&w, 0, 1);
    }
    bw_range(&w, 2, 2); /* BTYPE.DYNAMIC :28 */
    rc = deflate_dynamic_block(&w, in, n, processed, target, tokens);
    if (rc) break;
    processed += BLOCK_MAX;
    if (processed >= n) break;
  }
  free(tokens);
  if (rc) return rc;
  if (w.now_idx != 0) bw_range(&w, 0, (int)(8 - w.now_idx)); /* :35-37 */
  if (w.err) return w.err;
  *out_len = w.idx;
  return ZOR_OK;
}

/* The block loop of src/deflate.ts:20-34 over the blocks [start, start + len) of an n-byte input only: the raw bit
 * stream of that block range from bit 0 of `out`, BFINAL on its last block iff `final_range`, no padding beyond the
 * last byte; *out_bits = its length in bits.  Checker for the multi-GPU split of one buffer: the
 * ranges of a buffer, concatenated bit by bit, are the buffer's raw stream. */
int zor_deflate_range(const uint8_t* in, uint64_t n, uint64_t start, uint64_t len, int final_range, uint8_t* out, uint64_t cap,
                      uint64_t* out_bits) {
  const uint64_t heap = (len < BLOCK_MAX / 2) ? BLOCK_MAX : len * 2;
  if (cap < heap || (start % BLOCK_MAX) || start + len > n || len == 0) return ZOR_E_NOSPACE;
  memset(out, 0, heap);
  zor_bw w = {out, heap, 0, 0, 0, 0, 0};
  uint32_t* tokens = (uint32_t*)malloc(sizeof(uint32_t) * (BLOCK_MAX + 2));
  int rc = ZOR_OK;
  for (uint64_t processed = start; processed < start + len; processed += BLOCK_MAX) {
    const uint32_t target = (uint32_t)((start + len - processed < BLOCK_MAX) ? start + len - processed : BLOCK_MAX);
    bw_range(&w, (final_range && processed + BLOCK_MAX >= start + len) ? 1 : 0, 1);
    bw_range(&w, 2, 2);
    rc = deflate_dynamic_block(&w, in, n, processed, target, tokens);
    if (rc) break;
  }
  free(tokens);
  if (rc) return rc;
  if (w.err) return w.err;
  *out_bits = w.idx * 8 + w.now_idx;
  if (w.now_idx) out[w.idx] = (uint8_t)w.now_bits; /* the bits of the unfinished byte */
  return ZOR_OK;
}

/* zlib wrapper — src/zlib.ts:25-49 */
int zor_deflate(const uint8_t* in, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* out_len) {
  uint64_t need;
  zor_deflate_bound(n, &need);
  if (cap < need) return ZOR_E_NOSPACE;
  uint64_t raw_len = 0;
  int rc = zor_deflate_raw(in, n, out + 2, cap - 6, &raw_len);
  if (rc) return rc;
  out[0] = 0x78; /* CMF: CM=8 | CINFO=7<<4  :29-30 */
  out[1] = 0x9c; /* FLG: FCHECK=28 | FDICT=0<<5 | FLEVEL=2<<6  :32-34 */
  uint32_t a;
  zor_adler32(in, n, &a);
  out[2 + raw_len + 0] = (uint8_t)(a >> 24); /* :37-40 */
  out[2 + raw_len + 1] = (uint8_t)(a >> 16);
  out[2 + raw_len + 2] = (uint8_t)(a >> 8);
  out[2 + raw_len + 3] = (uint8_t)a;
  *out_len = raw_len + 6;
  return ZOR_OK;
}

/* ------------------------------------------------------------------------------------------
 * Bit reader — src/utils/BitReadStream.ts:1-50, restated state for state (the mix of the
 * eager readRange refill and read()'s isEnd flag decides which error a bad stream raises).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const uint8_t* buf;
  uint64_t len;
  int64_t idx;
  uint32_t now_bits;
  int now_len;
  int is_end;
  int err;
} zor_br;

static void br_init(zor_br* r, const uint8_t* buf, uint64_t len, uint64_t off) { /* :7-12 */
  r->buf = buf;
  r->len = len;
  r->idx = (int64_t)off;
  r->now_bits = off < len ? buf[off] : 0; /* undefined behaves as 0 in every use below */
  r->now_len = 8;
  r->is_end = 0;
  r->err = 0;
}
static int br_read(zor_br* r) { /* :14-32 */
  if (r->is_end) {
    r->err = ZOR_E_LACK;
    return 0;
  }
  int bit = (int)(r->now_bits & 1);
  if (r->now_len > 1) {
    r->now_len--;
    r->now_bits >>= 1;
  } else {
    r->idx++;
    if ((uint64_t)r->idx < r->len) {
      r->now_bits = r->buf[r->idx];
      r->now_len = 8;
    } else {
      r->now_len = 0;
      r->is_end = 1;
    }
  }
  return bit;
}
static uint32_t br_range(zor_br* r, int length) { /* :33-42 */
  while (r->now_len <= length) {
    r->idx++;
    uint32_t b = ((uint64_t)r->idx < r->len) ? r->buf[r->idx] : 0;
    r->now_bits |= b << r->now_len;
    r->now_len += 8;
  }
  uint32_t bits = r->now_bits & ((1u << length) - 1);
  r->now_bits >>= length;
  r->now_len -= length;
  return bits;
}
/* :43-49; length < 0 models Number.MAX_SAFE_INTEGER (empty table): loops until read() throws */
static uint32_t br_coded(zor_br* r, int length) {
  uint32_t bits = 0;
  if (length < 0) {
    while (!r->err) br_read(r);
    return 0;
  }
  for (int i = 0; i < length && !r->err; i++) bits = (bits << 1) | (uint32_t)br_read(r);
  return bits;
}

/* growable output — src/utils/Uint8WriteStream.ts:1-25 */
typedef struct {
  uint8_t* buf;
  uint64_t cap, idx;
  uint64_t limit; /* see the file header: more output than any stream of this length can hold */
  int runaway;
} zor_out;
static void out_write(zor_out* o, uint8_t v) {
  if (o->idx >= o->limit) {
    o->runaway = 1;
    return;
  }
  if (o->idx >= o->cap) {
    o->cap = o->cap ? o->cap * 2 : 65536;
    o->buf = (uint8_t*)realloc(o->buf, o->cap);
  }
  o->buf[o->idx++] = v;
}

/* canonical decode tables — src/huffman.ts:8-39: per bit length, codes first..first+count-1 map
 * to that length's symbols in ascending order. */
typedef struct {
  int lmin, lmax; /* lmin < 0: empty table (Number.MAX_SAFE_INTEGER in the reference) */
  uint32_t first[16];
  uint16_t count[16];
  uint16_t offs[16];
  uint16_t syms[400];
} zor_dtab;

static void dtab_build(zor_dtab* t, const uint8_t* lens, int nsym) {
  t->lmin = 99;
  t->lmax = 0;
  memset(t->count, 0, sizeof t->count);
  for (int s = 0; s < nsym; s++)
    if (lens[s]) {
      t->count[lens[s]]++;
      if (lens[s] < t->lmin) t->lmin = lens[s];
      if (lens[s] > t->lmax) t->lmax = lens[s];
    }
  if (t->lmax == 0) {
    t->lmin = -1;
    return;
  }
  uint32_t code = 0;
  uint16_t off = 0;
  for (int l = t->lmin; l <= t->lmax; l++) {
    t->first[l] = code;
    t->offs[l] = off;
    code += t->count[l];
    off = (uint16_t)(off + t->count[l]);
    code <<= 1;
  }
  uint16_t fill[16];
  memcpy(fill, t->offs, sizeof fill);
  for (int s = 0; s < nsym; s++)
    if (lens[s]) t->syms[fill[lens[s]]++] = (uint16_t)s;
}

/* the decode loop shared by src/inflate.ts:78-93, :158-171, :238-252, :267-281.
 * Returns symbol or -1 with r->err / *err set. */
static int dtab_decode(const zor_dtab* t, zor_br* r, int* err) {
  int cl = t->lmin;
  uint32_t code = br_coded(r, t->lmin);
  if (r->err) {
    *err = r->err;
    return -1;
  }
  while (1) {
    uint32_t rel = code - t->first[cl];
    if (code >= t->first[cl] && rel < t->count[cl]) return t->syms[t->offs[cl] + rel];
    if (t->lmax <= cl) {
      *err = ZOR_E_CORRUPT;
      return -1;
    }
    cl++;
    code = (code << 1) | (uint32_t)br_read(r);
    if (r->err) {
      *err = r->err;
      return -1;
    }
  }
}

/* the LZ77 copy shared by src/inflate.ts:94-116 and :253-290, including the reference's
 * behaviour on out-of-table codes (undefined base → NaN → zero bytes / no bytes). */
static int inflate_symbols(zor_br* r, zor_out* o, const zor_dtab* lt, const zor_dtab* dt, int fixed) {
  int err = 0;
  while (!r->is_end) {
    /* (file header) all-zero bits for 1 KiB past the end: the token loop has settled into a cycle that never ends */
    if (o->runaway || r->idx > (int64_t)r->len + 1024) return ZOR_E_LACK;
    int v = dtab_decode(lt, r, &err);
    if (v < 0) return err;
    if (v < 256) {
      out_write(o, (uint8_t)v);
      continue;
    }
    if (v == 256) break;
    int lc = v - 257;
    int have_len = lc < 29;
    uint32_t rl = have_len ? LEN_BASE[lc] : 0; /* undefined for codes 286/287 */
    if (have_len && LEN_XBITS[lc]) rl += br_range(r, LEN_XBITS[lc]);
    int dc;
    if (fixed) {
      dc = (int)br_coded(r, 5); /* :107 */
      if (r->err) return r->err;
    } else {
      dc = dtab_decode(dt, r, &err);
      if (dc < 0) return err;
    }
    int have_dist = dc < 30;
    uint32_t rd = have_dist ? DIST_BASE[dc] : 0;
    if (have_dist && DIST_XBITS[dc]) rd += br_range(r, DIST_XBITS[dc]);
    if (!have_len) continue; /* `i < undefined` is false: nothing copied */
    for (uint32_t i = 0; i < rl; i++) {
      /* NaN or negative source index reads `undefined`, stored as 0 */
      int64_t src = have_dist ? (int64_t)o->idx - (int64_t)rd : -1;
      out_write(o, (have_dist && src >= 0) ? o->buf[src] : 0);
    }
  }
  return ZOR_OK;
}

/* inflateDynamicBlock header — src/inflate.ts:120-204 */
static int inflate_dynamic(zor_br* r, zor_out* o) {
  const int HLIT = (int)br_range(r, 5) + 257;
  const int HDIST = (int)br_range(r, 5) + 1;
  const int HCLEN = (int)br_range(r, 4) + 4;
  uint8_t clens[19];
  memset(clens, 0, sizeof clens);
  for (int i = 0; i < HCLEN; i++) clens[CODELEN_ORDER[i]] = (uint8_t)br_range(r, 3); /* :127-136 */
  zor_dtab ct;
  dtab_build(&ct, clens, 19);
  uint8_t llens[288 + 8], dlens[64];
  memset(llens, 0, sizeof llens);
  memset(dlens, 0, sizeof dlens);
  int repeat = 0, codelen = 0, err = 0;
  const int total = HLIT + HDIST;
  for (int i = 0; i < total;) { /* :156-202 */
    int rc = dtab_decode(&ct, r, &err);
    if (rc < 0) return err;
    if (rc == 16) {
      repeat = 3 + (int)br_range(r, 2);
    } else if (rc == 17) {
      repeat = 3 + (int)br_range(r, 3);
      codelen = 0;
    } else if (rc == 18) {
      repeat = 11 + (int)br_range(r, 7);
      codelen = 0;
    } else {
      repeat = 1;
      codelen = rc;
    }
    if (codelen <= 0) {
      i += repeat;
    } else {
      while (repeat) { /* may run past `total`: extra symbols land in the distance table */
        if (i < HLIT) llens[i] = (uint8_t)codelen;
        else dlens[i - HLIT] = (uint8_t)codelen;
        i++;
        repeat--;
      }
    }
  }
  zor_dtab lt, dt;
  dtab_build(&lt, llens, 288);
  dtab_build(&dt, dlens, 64);
  return inflate_symbols(r, o, &lt, &dt, 0);
}

static int inflate_fixed(zor_br* r, zor_out* o) { /* src/inflate.ts:57-118, src/huffman.ts:41-53 */
  uint8_t llens[288];
  for (int i = 0; i <= 287; i++) llens[i] = (uint8_t)(i <= 143 ? 8 : i <= 255 ? 9 : i <= 279 ? 7 : 8);
  zor_dtab lt;
  dtab_build(&lt, llens, 288);
  return inflate_symbols(r, o, &lt, NULL, 1);
}

static int inflate_stored(zor_br* r, zor_out* o) { /* src/inflate.ts:42-55 */
  if (r->now_len < 8) br_range(r, r->now_len);
  uint32_t LEN = br_range(r, 8);
  LEN |= br_range(r, 8) << 8;
  uint32_t NLEN = br_range(r, 8);
  NLEN |= br_range(r, 8) << 8;
  if (LEN + NLEN != 65535) return ZOR_E_CORRUPT;
  for (uint32_t i = 0; i < LEN; i++) out_write(o, (uint8_t)br_range(r, 8));
  return ZOR_OK;
}

/* raw inflate from byte offset — src/inflate.ts:16-40.  *out is malloc'ed (free with zor_free). */
int zor_inflate_raw(const uint8_t* in, uint64_t c, uint64_t offset, uint8_t** out, uint64_t* out_len) {
  zor_out o = {NULL, 0, 0, 0, 0};
  o.limit = 1032ull * (offset < c ? c - offset : 0) + 65536;
  zor_br r;
  br_init(&r, in, c, offset);
  int bfinal = 0, rc = ZOR_OK;
  while (bfinal != 1) {
    bfinal = (int)br_range(&r, 1);
    int btype = (int)br_range(&r, 2);
    if (btype == 0) rc = inflate_stored(&r, &o);
    else if (btype == 1) rc = inflate_fixed(&r, &o);
    else if (btype == 2) rc = inflate_dynamic(&r, &o);
    else rc = ZOR_E_BTYPE3;
    if (!rc && o.runaway) rc = ZOR_E_LACK;
    if (rc) break;
    if (bfinal == 0 && r.is_end) { /* :34-36 */
      rc = ZOR_E_INSUFFICIENT;
      break;
    }
  }
  if (rc) {
    free(o.buf);
    *out = NULL;
    *out_len = 0;
    return rc;
  }
  if (!o.buf) o.buf = (uint8_t*)malloc(1);
  *out = o.buf;
  *out_len = o.idx;
  return ZOR_OK;
}

/* The same block loop (src/inflate.ts:22-37) as a map of the stream: the bit position of every block's first bit
 * and the output length behind every block — what the tests of the range decoder (zes_inflate_range_dev) and of
 * shard.inflate_split hold the engine's block starts against.  Not an interface of the reference. */
int zor_inflate_blocks(const uint8_t* in, uint64_t c, uint64_t offset, uint64_t* start_bit, uint64_t* out_end, uint32_t max,
                       uint32_t* count) {
  zor_out o = {NULL, 0, 0, 0, 0};
  o.limit = 1032ull * (offset < c ? c - offset : 0) + 65536;
  zor_br r;
  br_init(&r, in, c, offset);
  int bfinal = 0, rc = ZOR_OK;
  uint32_t k = 0;
  while (bfinal != 1 && !rc) {
    if (k < max) start_bit[k] = (uint64_t)(r.idx + 1) * 8 - (uint64_t)r.now_len;
    bfinal = (int)br_range(&r, 1);
    int btype = (int)br_range(&r, 2);
    if (btype == 0) rc = inflate_stored(&r, &o);
    else if (btype == 1) rc = inflate_fixed(&r, &o);
    else if (btype == 2) rc = inflate_dynamic(&r, &o);
    else rc = ZOR_E_BTYPE3;
    if (!rc && o.runaway) rc = ZOR_E_LACK;
    if (!rc && k < max) out_end[k] = o.idx;
    if (!rc) k++;
    if (!rc && bfinal == 0 && r.is_end) rc = ZOR_E_INSUFFICIENT;
  }
  free(o.buf);
  *count = k;
  return rc;
}

/* zlib wrapper — src/zlib.ts:11-23 */
int zor_inflate(const uint8_t* in, uint64_t c, uint8_t** out, uint64_t* out_len) {
  zor_br r;
  br_init(&r, in, c, 0);
  if (br_range(&r, 4) != 8) { /* CM :13-16; empty input reads undefined & 15 = 0 */
    *out = NULL;
    *out_len = 0;
    return ZOR_E_NOT_DEFLATE;
  }
  return zor_inflate_raw(in, c, 2, out, out_len);
}

void zor_free(void* p) { free(p); }
