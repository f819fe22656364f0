/*
 * zes_gen.c — deterministic, integer-only workload generators (host side).
 *
 * SURVEY.md App. B defines xorshift32 and lowent4k; `itext` is this repo's integer-only
 * stand-in for "enwik-like" text (enwik8 itself is not available offline).  The same three
 * generators are restated in tests/golden/make_golden.js so that fixtures made with the
 * reference under Node and buffers made here are byte-identical (pinned by sha256 in
 * tests/golden/manifest.json).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline uint32_t xs32(uint32_t* s) {
  uint32_t x = *s;
  x ^= x << 13;
  x ^= x >> 17;
  x ^= x << 5;
  *s = x;
  return x;
}

static void gen_xorshift(uint8_t* out, uint64_t n, uint32_t seed) {
  uint32_t s = seed;
  for (uint64_t i = 0; i < n; i++) out[i] = (uint8_t)(xs32(&s) & 0xff);
}

static void gen_lowent4k(uint8_t* out, uint64_t n, uint32_t seed) {
  uint8_t pat[4096];
  gen_xorshift(pat, 4096, seed);
  for (uint64_t i = 0; i < n; i++) out[i] = pat[i & 4095];
}

/* itext: Zipf-like word stream.  Vocabulary of 2048 words (1..9 letters, letters skewed
 * towards "etaoin…"), word index = product of four uniforms (heavy head), sentences of
 * 5..16 words, commas, full stops, occasional newlines, capitalised sentence starts. */
#define ITEXT_V 2048u
static void gen_itext(uint8_t* out, uint64_t n, uint32_t seed) {
  static const char letters[27] = "etaoinshrdlcumwfgypbvkjxqz";
  uint32_t s = seed ? seed : 1u;
  uint8_t (*words)[10] = (uint8_t (*)[10])malloc(ITEXT_V * 10);
  uint8_t* wlen = (uint8_t*)malloc(ITEXT_V);
  for (uint32_t w = 0; w < ITEXT_V; w++) {
    /* one draw per statement: C leaves operand evaluation order unspecified */
    uint32_t len = 1 + (xs32(&s) % 4u);
    len += xs32(&s) % 4u;
    len += xs32(&s) % 3u;
    wlen[w] = (uint8_t)len;
    for (uint32_t k = 0; k < len; k++) {
      uint32_t a = xs32(&s) % 26u, b = xs32(&s) % 26u;
      words[w][k] = (uint8_t)letters[(a * b) / 26u];
    }
  }
  uint64_t pos = 0;
  uint32_t left = 5 + xs32(&s) % 12u;
  int cap = 1;
  while (pos < n) {
    uint32_t a = xs32(&s) % ITEXT_V, b = xs32(&s) % ITEXT_V, c = xs32(&s) % ITEXT_V, d = xs32(&s) % ITEXT_V;
    uint32_t idx = (((((a * b) / ITEXT_V) * c) / ITEXT_V) * d) / ITEXT_V;
    for (uint32_t k = 0; k < wlen[idx] && pos < n; k++) {
      uint8_t ch = words[idx][k];
      if (cap && k == 0) ch = (uint8_t)(ch - 32);
      out[pos++] = ch;
    }
    cap = 0;
    left--;
    if (left == 0) {
      if (pos < n) out[pos++] = '.';
      uint32_t r = xs32(&s) % 4u;
      if (pos < n) out[pos++] = (r == 0) ? '\n' : ' ';
      cap = 1;
      left = 5 + xs32(&s) % 12u;
    } else {
      uint32_t r = xs32(&s) % 16u;
      if (r == 0 && pos < n) out[pos++] = ',';
      if (pos < n) out[pos++] = ' ';
    }
  }
  free(words);
  free(wlen);
}

int zes_gen(uint8_t* out, uint64_t n, uint32_t kind, uint32_t seed) {
  if (!out && n) return -18;
  switch (kind) {
    case 0: gen_xorshift(out, n, seed); return 0;
    case 1: gen_lowent4k(out, n, seed); return 0;
    case 2: gen_itext(out, n, seed); return 0;
    default: return -18;
  }
}
