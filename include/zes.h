/*
 * zes.h — C-ABI of the MI355X-native DEFLATE engine (drop-in for zlib.es's hot path).
 *
 * Every entry point replaces one interface of the reference (zprodev/zlib.es v0.6.0);
 * the citation after "replaces:" is the reference file:line the entry point stands in for.
 * Plain pointers and sizes only, no exceptions across the boundary: every function returns
 * ZES_OK (0) or a negative zes_status.  zes_strerror() returns, for the reference-defined
 * codes, the exact message string the reference throws, so a binding can rethrow it verbatim
 * (see INTEGRATION.md for the N-API / ctypes stubs).
 *
 * Two families:
 *   zes_*      — host pointers (what an FFI binding hands over); the library stages through
 *                its own device buffers (H2D, kernels, D2H).
 *   zes_*_dev  — device pointers (HBM-resident in/out); nothing crosses PCIe except a few
 *                scalars.  This is what bench.py times.
 *
 * The product path is HIP only.  There is no CPU fallback in this library: without a usable
 * gfx950 device every compute entry point returns ZES_E_DEVICE.
 */
#ifndef ZES_H
#define ZES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum zes_status {
  ZES_OK = 0,
  /* reference-defined errors (message strings identical to the reference's `throw new Error`) */
  ZES_E_NOT_DEFLATE = -1,   /* 'Not compressed by deflate'     src/zlib.ts:15               */
  ZES_E_BTYPE3 = -2,        /* 'Not supported BTYPE : 3'       src/inflate.ts:32            */
  ZES_E_CORRUPT = -3,       /* 'Data is corrupted'             src/inflate.ts:50,88,166,247,276; src/deflate.ts:172,190,202,216,224 */
  ZES_E_INSUFFICIENT = -4,  /* 'Data length is insufficient'   src/inflate.ts:35            */
  ZES_E_LACK = -5,          /* 'Lack of data length'           src/utils/BitReadStream.ts:15, BitWriteStream.ts:15 */
  /* engine-defined errors (no reference counterpart) */
  ZES_E_NOSPACE = -16,      /* caller's output capacity too small; *out_len holds the size needed when known */
  ZES_E_DEVICE = -17,       /* HIP runtime error / no gfx950 device */
  ZES_E_ARG = -18,          /* bad argument (null pointer, size overflow) */
  ZES_E_NOTRANGE = -19      /* zes_inflate_range_dev: the range does not hold a clean chain of reference-made blocks */
} zes_status;

/* Geometry of the reference format (src/const.ts:7). */
#define ZES_BLOCK_LEN 131072u

/* flags for zes_inflate*: */
#define ZES_F_DEFAULT 0u
#define ZES_F_NO_FASTPATH 1u   /* force the general (serial, any-stream) decoder: testing aid */
#define ZES_F_PIECES 4u        /* decode a reference-made stream piece by piece (1 MiB pieces) as streams of 512 MiB and more are
                                  * (256 MiB pieces): testing aid for that path; same results */
#define ZES_F_ALLOC_BOUND 8u   /* zes_inflate_alloc: the allocator may be asked EARLY for an upper estimate of the result's size (the
                                 result is then a prefix of what it returned: *out_len says how long), and a second time for the exact
                                 size if the estimate fell short — the last pointer it returned holds the result.  For callers whose
                                 memory can show a prefix (the N-API addon's pooled blocks): the download then runs beside the decode
                                 instead of behind it (64 MiB of random bytes: 18 -> 25 GiB/s).  The early request carries
                                 ZES_ALLOC_EARLY in its index argument; the allocator may answer it with NULL ("not now": no block of
                                 that size at hand) and is then asked once, later, for the exact size as without the flag */
#define ZES_ALLOC_EARLY 0x80000000u
#define ZES_F_LOOSE_CANDIDATES 2u /* block-start search without the reference's run-length-coding rules: more false
                                  * candidates reach the block decoder (testing aid for that path; same results) */

/* Exact reference message for a status (engine-defined codes get a descriptive string). */
const char* zes_strerror(int status);

/* Library/device lifecycle.  zes_init(device) binds the process to one HIP device (one process per GPU;
 * bench.py passes LOCAL_RANK).  Idempotent; a second call with another device returns ZES_E_ARG.  Every entry
 * point may be called from any thread: calls on one device are serialised by that device's lock and each makes the device
 * current on its calling thread for the duration of the call (HIP's current device is per thread).
 * replaces: nothing (the reference has no state); required because device scratch is pooled across calls. */
int zes_init(int device);
int zes_shutdown(void);
/* Gives the pooled device scratch back to the driver (every context's; ~10 bytes per input byte after a deflate call,
 * ~3 GB after another encoder's long stream) and keeps the contexts, streams and pinned staging: the next call
 * allocates what it needs again.  For a long-lived host that has had one large call.  replaces: nothing (the
 * reference's buffers are garbage collected). */
int zes_trim(void);
/* Bytes of pooled device scratch the library holds right now, over all contexts (what zes_trim would give back;
 * bench.py reports it per workload).  replaces: nothing. */
uint64_t zes_pool_bytes(void);
/* Several GPUs from ONE process (what a Node host is: SURVEY §8b `zes_init(int ngpus)`, "the batch API is where
 * multi-GPU concurrency lives").  zes_init_devices(n) gives the library n contexts, context i on device i (n <= 0: every
 * visible device) — its own stream, scratch pools, staging and lock each.  From then on the host-pointer entry points use
 * all of them: zes_deflate_batch / zes_inflate_batch_alloc partition their buffers by size (zes_partition) and run
 * every share on its own host thread against its own device, results straight into the caller's memory (the allocator
 * callback may then be called from several threads at once, always for distinct buffers); single host calls take the
 * devices in turn, so concurrent deflateAsync() calls land on different GPUs; device-pointer entry points run on the
 * device that holds their memory.  Results are identical to the one-device ones, buffer for buffer.  zes_init(device)
 * keeps meaning "this process drives that one device" (one process per GPU under torch.distributed).
 * zes_partition: owner[i] in [0, parts) for buffer i — longest first onto the lightest part so far, ties to the lower
 * index (the rule of zlib.es_amd/shard.py's partition()); no GPU involved.  zes_device_count: contexts in use.
 * replaces: the caller's own loop over buffers, README.md:28-42 (the reference is single-threaded). */
int zes_init_devices(int n);
int zes_device_count(void);
int zes_partition(const uint64_t* sizes, uint32_t count, uint32_t parts, uint32_t* owner);
/* Fills name (<= cap bytes) with the device's gcnArchName, *cus with its CU count. */
int zes_device_info(char* name, int cap, int* cus, uint64_t* hbm_bytes);

/* Page-locked host memory the DMA engines can read and write directly.  The host-pointer entry points accept any
 * memory (pageable buffers of 4 MiB and more are copied by the runtime's own call, shorter ones through a ring of pinned
 * chunks); buffers from here are handed to the DMA engines as they are, whatever their size.
 * A binding exposes it as an allocator for its callers' arrays (`allocPinned`).  Free before
 * zes_shutdown.  replaces: nothing (the reference works on ordinary Uint8Arrays). */
int zes_host_alloc(uint64_t n, void** p);
int zes_host_free(void* p);

/* Output capacity sufficient for zes_deflate of an n-byte input.
 * replaces: the `streamHeap` sizing in src/deflate.ts:16 (+6 for the zlib wrapper, src/zlib.ts:42). */
int zes_deflate_bound(uint64_t n, uint64_t* cap);

/* zlib-wrapped compress: out = 78 9C | raw deflate | Adler-32 BE.  Bit-exact with
 * replaces: `export function deflate(input)` src/zlib.ts:25-49 (→ src/deflate.ts:14-39, src/lz77.ts, src/huffman.ts:55-153, src/adler32.ts).
 * n == 0, n == 1 and n % 131072 == 1 return ZES_E_CORRUPT exactly as the reference throws.
 * Device forms (here and below): d_out, and in the batch forms every out_off, must be 16-byte aligned — results are
 * written as whole 16-byte groups — else ZES_E_ARG; the inflate device forms ask the same of d_in / in_off.  The
 * deflate device forms read an input at any alignment (the kernels fall back to narrower loads for an unaligned
 * block).  The host forms take any alignment. */
int zes_deflate(const uint8_t* in, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* out_len);
int zes_deflate_dev(const uint8_t* d_in, uint64_t n, uint8_t* d_out, uint64_t cap, uint64_t* out_len);

/* zlib-wrapped decompress.
 * replaces: `export function inflate(input)` src/zlib.ts:11-23 (→ src/inflate.ts:16-292, src/huffman.ts:8-53, src/utils/BitReadStream.ts).
 * Same accept-set and the same error for every malformed stream (checks only the CM nibble,
 * ignores FCHECK/FDICT and the Adler-32 trailer).  On ZES_E_NOSPACE *out_len = bytes needed. */
int zes_inflate(const uint8_t* in, uint64_t c, uint8_t* out, uint64_t cap, uint64_t* out_len, uint32_t flags);
int zes_inflate_dev(const uint8_t* d_in, uint64_t c, uint8_t* d_out, uint64_t cap, uint64_t* out_len, uint32_t flags);
/* Size-only pass (decodes, writes nothing to the caller, keeps nothing). */
int zes_inflate_size(const uint8_t* in, uint64_t c, uint64_t* n, uint32_t flags);
/* Decode once, then let the caller allocate the exact result: after the stream has been decoded on the device,
 * alloc(user, index, n) is called once, on the calling thread, for the n result bytes (n may be 0) and the bytes
 * are copied into what it returns; NULL from alloc → ZES_E_ARG.  The whole call holds the library's lock: there is
 * no state between calls.  index is 0 here (the batch form passes the buffer's index).
 * replaces: the growable Uint8WriteStream of src/inflate.ts:17,39 (src/utils/Uint8WriteStream.ts:1-25). */
typedef uint8_t* (*zes_alloc_fn)(void* user, uint32_t index, uint64_t n);
int zes_inflate_alloc(const uint8_t* in, uint64_t c, zes_alloc_fn alloc, void* user, uint64_t* out_len, uint32_t flags);

/* Raw DEFLATE, without the zlib wrapper, for callers that embed DEFLATE in another container.
 * zes_deflate_raw*  replaces: `export function deflate(input)` of src/deflate.ts:14-39 (what src/zlib.ts:35 wraps):
 *                   the same bytes as zes_deflate minus the 2-byte header and the 4-byte Adler-32 trailer.
 * zes_inflate_raw*  replaces: `export function inflate(input, offset = 0)` of src/inflate.ts:16-40: decodes the raw
 *                   stream that starts at byte `offset` of the c-byte buffer (bytes after the stream stay readable,
 *                   exactly as for the reference, whose zlib wrapper calls this with offset 2, src/zlib.ts:21).
 * Same statuses as the wrapped forms; there is no CM-nibble check on this path. */
int zes_deflate_raw(const uint8_t* in, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* out_len);
int zes_deflate_raw_dev(const uint8_t* d_in, uint64_t n, uint8_t* d_out, uint64_t cap, uint64_t* out_len);
int zes_inflate_raw(const uint8_t* in, uint64_t c, uint64_t offset, uint8_t* out, uint64_t cap, uint64_t* out_len, uint32_t flags);
int zes_inflate_raw_dev(const uint8_t* d_in, uint64_t c, uint64_t offset, uint8_t* d_out, uint64_t cap, uint64_t* out_len,
                        uint32_t flags);

/* Adler-32 of a buffer (standard value as an unsigned 32-bit).
 * replaces: `calcAdler32` src/adler32.ts:1-10 (byte extraction at src/zlib.ts:37-40). */
int zes_adler32(const uint8_t* in, uint64_t n, uint32_t* adler);
int zes_adler32_dev(const uint8_t* d_in, uint64_t n, uint32_t* adler);

/* Batch forms over independent buffers (configs 4/5 of BASELINE.json): count buffers, the
 * i-th at d_in + in_off[i] with in_len[i] bytes, written to d_out + out_off[i] (capacity
 * out_cap[i]); out_len[i] and status[i] filled per buffer.  All launches share the stream
 * so small buffers fill the chip together.  replaces: a caller's loop over deflate()/inflate()
 * (README.md:28-42) — the reference has no batch API. */
int zes_deflate_batch_dev(const uint8_t* d_in, const uint64_t* in_off, const uint64_t* in_len,
                          uint8_t* d_out, const uint64_t* out_off, const uint64_t* out_cap,
                          uint64_t* out_len, int32_t* status, uint32_t count);
int zes_inflate_batch_dev(const uint8_t* d_in, const uint64_t* in_off, const uint64_t* in_len,
                          uint8_t* d_out, const uint64_t* out_off, const uint64_t* out_cap,
                          uint64_t* out_len, int32_t* status, uint32_t count, uint32_t flags);
/* The same over host pointers (what a binding's deflateBatch(Uint8Array[]) / inflateBatch hands over): in[i] has
 * in_len[i] bytes.  Deflate writes buffer i to out[i] (capacity out_cap[i] >= zes_deflate_bound(in_len[i]));
 * inflate asks alloc(user, i, n) for buffer i's n result bytes once it is decoded (not called for a buffer whose
 * status is an error).  Per-buffer status[] / out_len[] as above; the return value is only non-zero when the call
 * as a whole could not run. */
int zes_deflate_batch(const uint8_t* const* in, const uint64_t* in_len, uint8_t* const* out, const uint64_t* out_cap,
                      uint64_t* out_len, int32_t* status, uint32_t count);
int zes_inflate_batch_alloc(const uint8_t* const* in, const uint64_t* in_len, zes_alloc_fn alloc, void* user,
                            uint64_t* out_len, int32_t* status, uint32_t count, uint32_t flags);

/* One buffer over several GPUs.  Blocks of the reference format are independent: every 131072-byte
 * block gets its own LZ77 index and its own Huffman codes, and blocks are concatenated bit by bit
 * (src/deflate.ts:20-37, src/lz77.ts:11-22).  So each GPU compresses a contiguous range of blocks and one of them
 * joins the bit streams; the result is bit-identical to zes_deflate of the whole buffer.
 * zes_deflate_range_dev: d_in = first byte of the range, n = its length (a multiple of 131072 unless final_range),
 *   n_readable >= n = bytes readable from d_in — the match finder compares up to 258 bytes past a block's end
 *   (src/lz77.ts:78-85), so a range that is not the last needs that much of the next one behind it.  final_range != 0
 *   sets BFINAL on the range's last block (src/deflate.ts:21-27).  d_out receives the raw bit stream from bit 0
 *   ((*out_bits + 7) / 8 bytes, zero padded; cap >= zes_deflate_bound(n)); *adler = Adler-32 of the range's bytes.
 * zes_deflate_join_dev: 78 9C | the pieces, bit-concatenated | zero pad | Adler-32 of the whole (combined from the
 *   pieces' values and lengths: src/adler32.ts:1-10 is associative in that sense), into d_out.  The pieces are device
 *   pointers on this GPU (4-byte aligned), in order; they are read, and d_out is written, in whole dwords: piece i
 *   must be readable up to the dword that holds its last bit (zes_deflate_range_dev's own output is), and
 *   cap >= the result's length rounded up to 4 — otherwise ZES_E_NOSPACE with *out_len = the result's length.
 * replaces: the block loop of src/deflate.ts:20-34 and the wrapper of src/zlib.ts:25-49, split at block boundaries. */
int zes_deflate_range_dev(const uint8_t* d_in, uint64_t n, uint64_t n_readable, int final_range, uint8_t* d_out, uint64_t cap,
                          uint64_t* out_bits, uint32_t* adler);
int zes_deflate_join_dev(const uint8_t* const* d_piece, const uint64_t* piece_bits, const uint32_t* piece_adler,
                         const uint64_t* piece_len, uint32_t count, uint8_t* d_out, uint64_t cap, uint64_t* out_len);

/* One stream over several GPUs, the other direction, and streams too long for one pass: a
 * reference-made stream is a chain of independent blocks of exactly 131072 output bytes (the last one shorter), so
 * any GPU can decode the blocks that START inside a range of the compressed bits once it has found them — the same
 * block-start search and block decoder as zes_inflate_dev, restricted to the range.
 *   d_in .. d_in + c     the piece: 16-byte aligned, c < 512 MiB, holding the range, what the range's last block needs
 *                        behind it (<= 144 KiB) and the header of the block after it
 *   lo_bit, own_bit      blocks that start at bit lo_bit <= s < own_bit (relative to d_in; lo_bit >= 16) belong to the call
 *   exact_start          != 0: a block starts exactly at lo_bit (the end bit of the piece before); 0: the first block
 *                        start found at or behind lo_bit begins the chain (a GPU that takes a middle part of the stream)
 *   d_out, cap           block k of the range goes to d_out + k * 131072
 *   *first_bit, *end_bit bit positions (relative to d_in) of the range's first block and behind its last one: consecutive
 *                        ranges fit when one's end is the next one's first; *nblocks, *final_block (BFINAL seen)
 * ZES_E_NOTRANGE: not a clean chain (another encoder's stream, a false block start): decode the stream with zes_inflate_dev.
 * replaces: the block loop of src/inflate.ts:22-37, split at block boundaries. */
int zes_inflate_range_dev(const uint8_t* d_in, uint64_t c, uint64_t lo_bit, uint64_t own_bit, int exact_start, uint8_t* d_out,
                          uint64_t cap, uint64_t* out_len, uint64_t* first_bit, uint64_t* end_bit, uint32_t* nblocks, int* final_block);

/* Stage-level entry points (device pointers) used by the kernel parity tests; each mirrors
 * one internal function of the reference. */
/* replaces: generateLZ77Codes src/lz77.ts:24-119 for the block [start, start+len) of an n-byte input.
 * tokens[i] = literal byte, or 0x80000000 | (len-3) << 16 | (dist-1) for a match. */
int zes_stage_lz77_dev(const uint8_t* d_in, uint64_t n, uint64_t start, uint32_t len,
                       uint32_t* h_tokens, uint32_t* ntokens);
/* replaces: the code-length half of generateDeflateHuffmanTable src/huffman.ts:55-115.
 * hist[nsym] symbol counts → lens[nsym] code lengths (0 = unused), limit maxlen (15 or 7). */
int zes_stage_huff_lengths_dev(const uint32_t* h_hist, uint32_t nsym, uint32_t maxlen, uint8_t* h_lens);

/* Checks, on the device this context drives, the hardware behaviour k_lz_sort's stable ranks rest on: lanes of one
 * wavefront whose returning LDS add (ds_add_rtn_u32) meets in one word receive their old values in ascending lane order.
 * 256 workgroups x 16 wavefronts x iters rounds x 4 adds over six digit patterns; *bad = values that differ from the rank
 * computed with ballots (0 on gfx950), *checked = values compared.  A diagnostic for tests/test_gpu_hw_props.py.
 * replaces: nothing. */
int zes_selftest_lds_order(uint32_t iters, uint32_t seed, uint64_t* bad, uint64_t* checked);

/* Timing of the last *_dev call's kernels, measured with HIP events on the library's own
 * stream: name/ms pairs for bench.py's roofline leg.  Returns the number of entries. */
typedef struct zes_ktime { const char* name; float ms; uint32_t launches; } zes_ktime;
int zes_last_kernel_times(zes_ktime* out, int cap);
/* Which decoder produced the last zes_inflate*() result: 1 block-parallel (reference-made streams),
 * 2 segment-parallel (any valid stream), 3 sequential wavefront, 4 exact single-lane restatement
 * (DESIGN.md §4); 0 if the call failed before decoding. */
int zes_last_inflate_tier(void);
/* (After zes_init_devices: zes_last_inflate_tier / zes_last_kernel_times report on the context that served the calling
 * thread's last call; zes_set_profiling switches every context.) */
int zes_set_profiling(int on);

/* Deterministic integer-only workload generators: host side, used by bench.py,
 * the tests and the JS fixture script alike. kind: 0 xorshift32 bytes, 1 lowent4k, 2 itext. */
int zes_gen(uint8_t* out, uint64_t n, uint32_t kind, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif /* ZES_H */
