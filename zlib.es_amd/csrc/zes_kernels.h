// zes_kernels.h — launch geometry and kernel declarations shared by the .hip files and the host API.
#pragma once
#include "zes_common.h"

#define SORT_THREADS 1024
#define SORT_WAVES (SORT_THREADS / 64)
#ifndef MATCH_THREADS
#define MATCH_THREADS 1024
#endif
#define IDX_THREADS 1024
// k_lz_sort / k_lz_index -> the match finders, per block in idx_a[g][ZES_BLK-1]: kept positions | flags
#define ZES_SORT_LAZY 0x80000000u   // the block's index is sd[] / inv[]: it belongs to k_lz_match_lazy
#define ZES_SORT_INDEX 0x40000000u  // a dense block k_lz_sort left to k_lz_index
#define ZES_SORT_REDO 0x20000000u   // k_lz_index handed the block back (a class too large for its LDS): k_lz_sort, second launch
#define ZES_INV_NONE 0xFFFFFFFFu    // inv entry of a position without a candidate
#define PARSE_THREADS 1024
#define EMIT_THREADS 1024
#define HUFF_THREADS_HOST 256
#define ADLER_THREADS 256
#define ADLER_CHUNK 65536u
#define ZES_PAR_DBG_ROW 32  // u64 slots per work item of k_inf_block_par's ZES_DEBUG_PHASES stamps
// k_lz_match_lazy -> k_lz_parse, per block: [0] = 1 when the mask is there, [4..] a bit per position of the greedy chain
#define ZES_TMASK_WORDS (131072 / 32 + 4)
// k_lz_match -> k_lz_parse, per block: [0] = matches found (all ones: a block of k_lz_match_lazy), [1..] the first ZES_MLIST_CAP of them as position | (length - 3) << 17
#define ZES_MLIST_CAP 4095u
#define ZES_MLIST_WORDS 4096u
#define PAR_THREADS 1024
#define PAR_WAVES (PAR_THREADS / 64)
#define INF_SCAN_THREADS 256
#ifndef INF_SCAN_BYTES
#define INF_SCAN_BYTES 8192u
#endif
#define SEG_BUCKETS 2048u  // T2: at most this many candidate block starts are kept (one per bucket of the compressed stream)

// One entry per buffer of an inflate call (T1); entry [nbuf] is a sentinel carrying the totals.
struct ZesInfBuf {
  uint64_t in_off, c, out_off, cap;
  uint32_t first_chunk;  // k_inf_scan: first workgroup of this buffer
  uint32_t cand_base;    // start of the buffer's region in cand[] / cand_sorted[] / cres[] / map[]
  uint32_t cand_cap;     // entries in that region
  uint32_t work_first;   // k_inf_block_par: first work item of this buffer
  // range form (zes_inflate_range_dev: one piece of a longer stream): the chain of blocks starts at candidate value
  // start_rel (bit position - 16; 0 = right behind the zlib header) or, with ZES_START_ANY, at the first candidate at
  // or behind it; candidates at or behind own_rel only serve as end estimates and are left to the next piece
  uint32_t start_rel;
  uint32_t own_rel;      // 0xFFFFFFFF: the whole buffer
  uint32_t range_flags;  // ZES_START_ANY
  uint32_t pad;
};
#define ZES_START_ANY 1u

// work_first of the table's sentinel entry in one-buffer calls: the decode kernels take the number of work
// items from the candidate counter on the device (the sentinel's cand_cap holds the launch bound)
#define ZES_WORK_AUTO 0xFFFFFFFFu

struct ZesCandRes {
  uint64_t end_bit;   // absolute bit just past the block's EOB
  uint32_t out_len;
  uint32_t flags;     // bit0 ok, bit1 bfinal
};

// k_inf_block_par*, one-buffer calls: where the results also go, in the host's page-locked memory (the host then checks
// the chain of blocks itself; all null: nothing)
struct ZesParMirror {
  ZesCandRes* cres_host;        // [work] a copy of every work item's result ...
  uint32_t* start_host;         // ... and the bit its block starts at
  const uint32_t* counters;     // the search's counters (final when this kernel starts), copied by work item 0 ...
  uint32_t* counters_host;      // ... to here
  uint32_t counter_words;
};

// T2 (segment-parallel inflate): result of one work item of k_inf_seg_scan
struct ZesSegRes {
  uint64_t end_bit;   // absolute bit where the segment stopped
  uint64_t out_len;   // bytes the segment produces
  uint32_t flags;     // bit0 ok, bit1 the segment ends with the final block, bit2 its symbols did not fit the symbol store
  uint32_t next;      // work item that starts at end_bit (0 = none)
};

// one stored block of a stream that consists of stored blocks only
struct ZesStoredBlk {
  uint64_t src, dst;  // byte offsets inside the stream / inside the output
  uint32_t len, pad;
};

// T2: one entry per buffer of a group; the per-work-item arrays (sres, maps, order, seg, prefix, wins) are laid out
// buffer after buffer, a buffer's part starting at index work_first
struct ZesSegJob {
  uint64_t in_off, c;
  uint64_t sym_base;   // first dword of the buffer's share of the symbol store
  uint32_t cand_base;  // the buffer's candidates in the sorted list
  uint32_t ncand;
  uint32_t work_first;
  uint32_t nseg;       // length of the buffer's chain (0 = not decoded by this tier): k_inf_seg_windows
  // a piece of a longer stream (streams of 512 MiB and more go through this tier piece by piece): work item 0 starts
  // at bit start0 of the piece (16: right behind the zlib header), and a chain that ends in front of a block the piece
  // does not hold whole is accepted as far as it got (ZES_SEG_PARTIAL)
  uint32_t start0;
  uint32_t flags;
};
#define ZES_SEG_PARTIAL 1u

// where a buffer of a segment-parallel group goes (k_inf_seg_translate)
struct ZesSegOut {
  uint64_t out_off, cap;
  uint32_t nseg;  // segments to translate (0: none)
  uint32_t hist;  // bytes of output in front of out_off that exist (a later piece of a long stream: up to 32768)
};

#ifdef __HIPCC__
// inflate direction (zes_inflate.hip)
__global__ void k_inf_first_bytes(const uint8_t*, const uint64_t*, uint8_t*, uint32_t);
__global__ void k_inf_scan(const uint8_t*, const ZesInfBuf*, uint32_t, unsigned long long*, uint32_t, uint32_t*, uint8_t*, uint32_t,
                           const uint8_t*);
__global__ void k_inf_set_table1(ZesInfBuf, ZesInfBuf, ZesInfBuf*, uint32_t*, uint32_t);
__global__ void k_inf_verify(const uint8_t*, const ZesInfBuf*, const unsigned long long*, uint32_t, uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t*, uint32_t, uint32_t);
__global__ void k_inf_verify_long(const uint8_t*, const ZesInfBuf*, const unsigned long long*, uint32_t, uint32_t*, uint32_t*, uint32_t*, uint32_t, const uint32_t*, uint32_t);
__global__ void k_inf_ranksort(const ZesInfBuf*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t);
__global__ void k_inf_decode(const uint8_t*, uint8_t*, const ZesInfBuf*, ZesRes*, uint64_t*);
__global__ void k_inf_stored_walk(const uint8_t*, uint64_t, uint64_t, uint64_t, ZesStoredBlk*, ZesRes*);
__global__ void k_inf_stored_copy(const uint8_t*, uint64_t, uint8_t*, uint64_t, const ZesStoredBlk*);
__global__ void k_inf_stored_find(const uint8_t*, uint64_t, uint32_t, uint32_t*, uint32_t*);
__global__ void k_inf_stored_rank(const uint8_t*, uint64_t, uint32_t, const uint32_t*, const uint32_t*, uint32_t, uint64_t, ZesStoredBlk*, ZesRes*);
__global__ void k_inf_cand_bucket(const uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*);
__global__ void k_inf_cand_compact(const uint32_t*, uint32_t*, uint32_t*);
__global__ void k_inf_seg_order(const ZesSegJob*, const uint32_t*, uint32_t*);
__global__ void k_inf_seg_scan(const uint8_t*, const ZesSegJob*, uint32_t, const uint32_t*, ZesSegRes*, uint32_t*, uint32_t*, uint32_t,
                               const uint32_t*, uint32_t*, const uint32_t*, uint64_t*);
__global__ void k_inf_seg_scan_short(const uint8_t*, const ZesSegJob*, uint32_t, const uint32_t*, ZesSegRes*, uint32_t*, uint32_t*, uint32_t,
                                     const uint32_t*, uint32_t*, const uint32_t*, uint64_t*);
__global__ void k_inf_seg_block_par(const uint8_t*, const ZesSegJob*, uint32_t, const uint32_t*, ZesSegRes*, uint32_t*, uint32_t*, uint32_t, uint32_t*,
                                    unsigned long long*, uint64_t, uint64_t, uint64_t*, unsigned long long*);
__global__ void k_inf_seg_chain(const ZesSegJob*, const ZesSegRes*, uint32_t*, uint64_t*, ZesRes*, uint32_t*);
__global__ void k_inf_seg_translate(uint8_t*, const ZesSegJob*, const ZesSegOut*, const uint32_t*, const ZesSegRes*, const uint32_t*, const uint64_t*,
                                    const uint8_t*, const uint32_t*, const uint64_t*, uint32_t*);
#define SEGWIN_GROUP 32u
__global__ void k_inf_seg_win_group(const uint32_t*, const uint32_t*, const ZesSegJob*, uint32_t*);
__global__ void k_inf_seg_win_top(const uint32_t*, const ZesSegJob*, uint8_t*, const uint8_t*, const ZesSegOut*);
__global__ void k_inf_seg_win_fin(const uint32_t*, const ZesSegJob*, const uint8_t*, uint8_t*);
__global__ void k_inf_seg_decode(const uint8_t*, uint64_t, uint64_t, uint8_t*, uint64_t, uint64_t, const uint32_t*, const ZesSegRes*,
                                 const uint32_t*, const uint64_t*, const uint8_t*, uint32_t*, uint32_t, uint32_t, uint32_t);
__global__ void k_inf_block_par(const uint8_t*, uint8_t*, const ZesInfBuf*, uint32_t, const uint32_t*, const uint32_t*, const uint32_t*,
                                ZesCandRes*, unsigned long long*, const uint32_t*, const uint32_t*, uint32_t*, ZesParMirror);
__global__ void k_inf_block_par2(const uint8_t*, uint8_t*, const ZesInfBuf*, uint32_t, const uint32_t*, const uint32_t*, const uint32_t*,
                                ZesCandRes*, unsigned long long*, const uint32_t*, const uint32_t*, uint32_t*, ZesParMirror);
__global__ void k_inf_move_slots(uint8_t*, const uint8_t*, const uint32_t*, const uint32_t*, uint32_t);
__global__ void k_inf_chain(const ZesInfBuf*, const uint32_t*, const uint32_t*, const ZesCandRes*, const uint32_t*, uint32_t*, ZesRes*, const uint32_t*, uint32_t, uint32_t*);
__global__ void k_inf_chain_range(const ZesInfBuf*, const uint32_t*, const uint32_t*, const ZesCandRes*, ZesRes*, unsigned long long*);
__global__ void k_inf_set_table_range(ZesInfBuf, ZesInfBuf, ZesInfBuf*, uint32_t*, uint32_t, const unsigned long long*, unsigned long long);
__global__ void k_inf_exact(const uint8_t*, uint64_t, uint64_t, uint8_t*, uint64_t, uint64_t, const uint64_t*, ZesRes*);
// deflate direction (zes_deflate.hip)
void zes_sort_set_dbg(unsigned long long*);
#ifdef WD_PROFILE
void zes_wd_set_dbg(unsigned long long*);
#endif
void zes_parse_set_dbg(unsigned long long*);
void zes_lazy_set_dbg(unsigned long long*);
void zes_huff_set_dbg(unsigned long long*);
__global__ void k_make_blks(ZesBuf, uint32_t, ZesBuf*, ZesBlk*, uint32_t, unsigned long long*);
__global__ void k_lz_sort(const uint8_t*, const ZesBuf*, const ZesBlk*, uint32_t*, uint32_t*, uint32_t*, uint16_t*, uint32_t);
__global__ void k_lz_index(const uint8_t*, const ZesBuf*, const ZesBlk*, uint32_t*, uint32_t*, uint32_t*, uint16_t*);
#define ZES_SORT_MODE_FIRST 0u   // k_lz_sort: every block; dense ones are left to k_lz_index when bit 8 is set
#define ZES_SORT_MODE_REDO 1u    // k_lz_sort: only the blocks k_lz_index handed back
#define ZES_SORT_USE_INDEX 256u
__global__ void k_lz_match_lazy(const uint8_t*, const ZesBuf*, const ZesBlk*, const uint32_t*, const uint32_t*, const uint16_t*, uint32_t*, uint32_t*, uint32_t*,
                                const uint32_t*);
__global__ void k_lz_order(const uint32_t*, uint32_t, uint32_t*);
__global__ void k_lz_match(const uint8_t*, const ZesBuf*, const ZesBlk*, const uint32_t*, uint32_t*, uint32_t*);
__global__ void k_lz_parse(const uint8_t*, const ZesBuf*, ZesBlk*, const uint32_t*, uint32_t*, uint32_t*, const uint32_t*, const uint32_t*);
__global__ void k_lz_parse_small(const uint8_t*, const ZesBuf*, ZesBlk*, const uint32_t*, uint32_t*, uint32_t*, const uint32_t*, const uint32_t*);
__global__ void k_huff(ZesBlk*, const uint32_t*, uint32_t*, uint32_t*);
__global__ void k_huff_lengths_only(const uint32_t*, uint32_t, uint32_t, uint8_t*);
__global__ void k_selftest_lds_order(unsigned long long*, uint32_t, uint32_t);
__global__ void k_adler(const uint8_t*, uint64_t, uint64_t, unsigned long long*);
__global__ void k_adler_blocks(const uint8_t*, const ZesBuf*, const ZesBlk*, unsigned long long*);
__global__ void k_layout(uint8_t*, const ZesBuf*, ZesBlk*, const unsigned long long*, ZesRes*);
__global__ void k_emit(uint8_t*, const ZesBuf*, const ZesBlk*, const uint32_t*, const uint32_t*, const uint32_t*);
__global__ void k_zero_u64(unsigned long long*, uint32_t);
__global__ void k_bits_place(uint32_t*, uint64_t, const uint32_t*, uint64_t);
#endif
