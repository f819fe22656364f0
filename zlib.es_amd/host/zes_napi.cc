// zes_napi.cc — N-API addon: the binding between the TypeScript façade (zlib.ts) and the C-ABI of
// include/zes.h.  Thin on purpose: argument marshalling and error translation only; every byte
// of work happens in libzes_hip.so (HIP kernels).  Synchronous like the reference's functions
// (src/zlib.ts:11,25); deflateAsync / inflateAsync run the same calls on the libuv thread pool and return
// Promises (the JS thread stays free while a GPU works).  N-API version 3 calls only (Node >= 10).
#include <node_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <csignal>
#include <execinfo.h>
#include <unistd.h>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/zes.h"

namespace {

napi_value throw_status(napi_env env, int status) {
  // plain `Error` with the reference's exact message (src/zlib.ts:15, src/inflate.ts:32,35,50,...)
  napi_throw_error(env, nullptr, zes_strerror(status));
  return nullptr;
}

bool get_bytes(napi_env env, napi_value v, const uint8_t** data, size_t* len) {
  bool is_ta = false;
  if (napi_is_typedarray(env, v, &is_ta) == napi_ok && is_ta) {
    napi_typedarray_type type;
    void* p = nullptr;
    size_t n = 0;
    napi_value ab;
    size_t off;
    if (napi_get_typedarray_info(env, v, &type, &n, &p, &ab, &off) != napi_ok) return false;
    if (type != napi_uint8_array && type != napi_uint8_clamped_array && type != napi_int8_array) return false;
    *data = static_cast<const uint8_t*>(p);
    *len = n;
    return true;
  }
  bool is_buf = false;
  if (napi_is_buffer(env, v, &is_buf) == napi_ok && is_buf) {
    void* p = nullptr;
    size_t n = 0;
    if (napi_get_buffer_info(env, v, &p, &n) != napi_ok) return false;
    *data = static_cast<const uint8_t*>(p);
    *len = n;
    return true;
  }
  return false;
}

// ---- result memory ----
// A result is a fresh Uint8Array with its own exact-length ArrayBuffer (src/zlib.ts:42).  Fresh memory is what a large call
// pays most for on the host side: 64 MiB of untouched pages are 16 384 page faults under the copy that fills them, ~30 ms
// where the GPU's work and both trips over PCIe take 2.5.  So results of 1 MiB and more live in blocks of page-locked memory
// (zes_host_alloc: no faults, and the copy engines write them directly) that come back to a small pool when the ArrayBuffer is
// collected and are handed out again.  Node runs the finalizers of external ArrayBuffers between event-loop turns, not
// inside a synchronous loop: a host that awaits (deflateAsync / inflateAsync, any server) gets its blocks back and runs at
// the library's pace; a tight synchronous loop over 64 MiB calls gets fresh blocks until BIG_OUTSTANDING_MAX are out, then
// plain malloc as before.  trim() empties the pool.
// Set when the last environment (or the process) goes down: from then on nothing here calls into the library or the HIP
// runtime any more — page-locked blocks are left to the process's end.
std::atomic<bool> g_exiting{false};
std::atomic<long> g_dbg_hit{0}, g_dbg_miss{0}, g_dbg_released{0}, g_dbg_fresh{0};
std::atomic<int> g_envs{0};
void mark_exiting_atexit() { g_exiting.store(true); }

constexpr size_t BIG_MIN = 1u << 20;                 // smaller results: malloc, as before
constexpr size_t BIG_KEEP_MAX = 768u << 20;          // bytes the pool keeps for reuse
constexpr size_t BIG_OUTSTANDING_MAX = 1024u << 20;  // (V8 collects a few calls behind: 64 MiB calls in an awaited loop hold ~0.5 GiB at any time)  // pooled bytes in the hands of JS beyond which new results are not pooled
struct BigPool {
  struct Blk {
    uint8_t* p;
    size_t cap;
  };
  std::mutex mu;
  std::vector<Blk> free_;
  size_t kept = 0, out = 0;
  size_t want = 0;  // a request the pool could not serve from inside the library (see take): the next top_up() gets such a block
  // a block of at least n bytes, or nullptr (the caller falls back to malloc).  fresh = false: from inside one of the
  // library's allocator callbacks — they run under the library's lock, and a new page-locked block comes from the library
  // (zes_host_alloc takes that lock): only what the pool holds; the miss is noted and made good after the call.
  uint8_t* take(size_t n, size_t* cap, bool fresh, bool note_miss = true) {
    {
      std::lock_guard<std::mutex> lk(mu);
      size_t best = free_.size();
      for (size_t i = 0; i < free_.size(); i++)
        if (free_[i].cap >= n && free_[i].cap <= 2 * n + (8u << 20) && (best == free_.size() || free_[i].cap < free_[best].cap)) best = i;
      if (best != free_.size()) {
        g_dbg_hit++;
        Blk b = free_[best];
        free_.erase(free_.begin() + (long)best);
        kept -= b.cap;
        out += b.cap;
        *cap = b.cap;
        return b.p;
      }
      g_dbg_miss++;
      if (out + n > BIG_OUTSTANDING_MAX) return nullptr;
      if (!fresh) {
        if (note_miss) want = n > want ? n : want;
        return nullptr;
      }
    }
    void* p = nullptr;
    const size_t want = (n + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
    if (zes_host_alloc(want, &p) != 0 || !p) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    out += want;
    *cap = want;
    return static_cast<uint8_t*>(p);
  }
  void give(uint8_t* p, size_t cap) {
    if (g_exiting.load()) return;
    {
      std::lock_guard<std::mutex> lk(mu);
      out -= cap < out ? cap : out;
      if (kept + cap <= BIG_KEEP_MAX) {
        free_.push_back({p, cap});
        kept += cap;
        return;
      }
    }
    zes_host_free(p);
  }
  // after a library call whose callback missed: one block of the size it asked for, for the next call of that kind
  void top_up() {
    size_t n = 0;
    {
      std::lock_guard<std::mutex> lk(mu);
      n = want;
      want = 0;
      if (!n || out + kept + n > BIG_OUTSTANDING_MAX) return;
    }
    void* p = nullptr;
    const size_t cap = (n + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
    if (zes_host_alloc(cap, &p) != 0 || !p) return;
    std::lock_guard<std::mutex> lk(mu);
    if (kept + cap <= BIG_KEEP_MAX) {
      free_.push_back({static_cast<uint8_t*>(p), cap});
      kept += cap;
      return;
    }
    zes_host_free(p);  // (cannot happen under the limits above; never keep more than the budget)
  }
  void clear() {
    if (g_exiting.load()) return;
    std::vector<Blk> all;
    {
      std::lock_guard<std::mutex> lk(mu);
      all.swap(free_);
      kept = 0;
    }
    for (Blk& b : all) zes_host_free(b.p);
  }
};
BigPool& g_big = *new BigPool;  // (never destroyed: an environment's cleanup may still come by during exit())

// memory for a result of up to n bytes: pooled (cap > 0) or malloc'd (cap == 0)
struct ResultMem {
  uint8_t* p = nullptr;
  size_t cap = 0;
};
ResultMem result_alloc(size_t n, bool fresh = true) {
  ResultMem m;
  if (n >= BIG_MIN) m.p = g_big.take(n, &m.cap, fresh);
  if (!m.p) {
    m.cap = 0;
    m.p = static_cast<uint8_t*>(malloc(n ? n : 1));
  }
  return m;
}
void result_free(ResultMem m) {
  if (!m.p) return;
  if (m.cap)
    g_big.give(m.p, m.cap);
  else
    free(m.p);
}
// ---- whose memory is still in use: no finalizers ----
// The memory behind an array handed to JS (a result, an allocPinned() array) is an external ArrayBuffer WITHOUT a finalizer;
// the addon keeps a weak reference to it and looks, at the start of every call, which of them the collector has cleared:
// those blocks go back (to the pool, to the library, to free()).  Why not a finalizer: Node 12 hands an N-API finalizer to
// the environment's immediate queue without keeping the N-API environment alive, frees that environment in a cleanup hook
// when the process ends, and runs the queue afterwards — v8::HandleScope on freed memory: seven of ten bench_host.js runs
// ended in SIGSEGV behind their last line (any array alive at exit has its finalizer queued there).  And a finalizer only
// ever runs between two turns of the event loop, so a synchronous loop over deflate() got fresh memory for every result,
// page by page; a weak reference is cleared by the collection itself, and the next call of the loop has the block again.
struct Track {
  napi_ref ref;   // weak: to the ArrayBuffer
  uint8_t* p;
  size_t cap;     // pooled block: its capacity; 0: malloc'd (results) — or page-locked by allocPinned (pinned)
  int64_t told;   // bytes reported with napi_adjust_external_memory
  bool pinned;    // allocPinned(): goes back with zes_host_free
};
struct EnvTracks {
  napi_env env;
  std::vector<Track> v;
  size_t cursor = 0;
};
std::mutex g_tracks_mu;
std::vector<EnvTracks*> g_tracks;  // one per environment (a Worker has its own); an entry is used by its environment's thread only
EnvTracks* tracks_of(napi_env env) {
  std::lock_guard<std::mutex> lk(g_tracks_mu);
  for (EnvTracks* t : g_tracks)
    if (t->env == env) return t;
  return nullptr;
}
void release_memory(const Track& t) {
  if (t.pinned) {
    if (!g_exiting.load()) zes_host_free(t.p);
  } else {
    ResultMem m;
    m.p = t.p;
    m.cap = t.cap;
    result_free(m);
  }
}
constexpr size_t SWEEP_MAX = 4096;  // references looked at per call (a host that keeps more arrays alive is swept in turns)
void sweep(napi_env env) {
  EnvTracks* et = tracks_of(env);
  if (!et || et->v.empty()) return;
  size_t todo = et->v.size() < SWEEP_MAX ? et->v.size() : SWEEP_MAX;
  size_t i = et->cursor < et->v.size() ? et->cursor : 0;
  while (todo--) {
    if (i >= et->v.size()) i = 0;
    if (et->v.empty()) break;
    napi_value val = nullptr;
    if (napi_get_reference_value(env, et->v[i].ref, &val) == napi_ok && val == nullptr) {
      const Track t = et->v[i];
      et->v[i] = et->v.back();
      et->v.pop_back();
      napi_delete_reference(env, t.ref);
      if (t.told) {
        int64_t now = 0;
        napi_adjust_external_memory(env, -t.told, &now);
      }
      release_memory(t);
      g_dbg_released++;
    } else {
      i++;
    }
  }
  et->cursor = i;
}
// the environment goes down: its arrays with it (no N-API call from here; pooled blocks stay pooled for other environments)
void env_cleanup(void* arg) {
  napi_env env = static_cast<napi_env>(arg);
  EnvTracks* mine = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_tracks_mu);
    for (size_t i = 0; i < g_tracks.size(); i++)
      if (g_tracks[i]->env == env) {
        mine = g_tracks[i];
        g_tracks.erase(g_tracks.begin() + (long)i);
        break;
      }
  }
  if (getenv("ZES_NAPI_DEBUG"))
    fprintf(stderr, "zes_napi: pool hits %ld misses %ld, arrays released by sweeps %ld, still tracked %zu\n", g_dbg_hit.load(), g_dbg_miss.load(),
            g_dbg_released.load(), mine ? mine->v.size() : 0);
  const bool last = g_envs.fetch_sub(1) == 1;
  if (last) g_exiting.store(true);
  if (mine) {
    if (!last)
      for (const Track& t : mine->v) release_memory(t);
    delete mine;
  }
}
// hands p[0, n) to JS as the ArrayBuffer of a fresh Uint8Array and starts tracking it; on failure the memory is released
napi_value adopt(napi_env env, uint8_t* p, size_t n, size_t cap, bool pinned) {
  Track t;
  t.ref = nullptr;
  t.p = p;
  t.cap = cap;
  t.told = (int64_t)(cap ? cap : n);
  t.pinned = pinned;
  EnvTracks* et = tracks_of(env);
  napi_value ab, ta;
  if (!et || napi_create_external_arraybuffer(env, p, n, nullptr, nullptr, &ab) != napi_ok) {
    release_memory(t);
    return nullptr;
  }
  if (napi_create_reference(env, ab, 0, &t.ref) != napi_ok) {  // (the ArrayBuffer is garbage from here; its memory must outlive it: kept)
    napi_throw_error(env, nullptr, "zes: out of memory");
    return nullptr;
  }
  int64_t now = 0;
  if (napi_adjust_external_memory(env, t.told, &now) != napi_ok) t.told = 0;
  et->v.push_back(t);
  if (napi_create_typedarray(env, napi_uint8_array, n, ab, 0, &ta) != napi_ok) return nullptr;  // (tracked: released once collected)
  return ta;
}
// the result's memory becomes its ArrayBuffer (exact length: `buffer.byteLength === length` like src/zlib.ts:42; a pooled
// block is longer than what the ArrayBuffer shows of it): no second copy of the result on the JS thread.  Takes the
// memory over (gives it back on failure).  V8 is told how much memory hangs on the object — it does not count an external
// ArrayBuffer's bytes by itself — so that a loop over large calls makes it collect, and the blocks come back to the pool.
napi_value take_u8(napi_env env, ResultMem m, size_t n) {
  if (!m.cap) {
    void* shrunk = realloc(m.p, n ? n : 1);
    if (shrunk) m.p = static_cast<uint8_t*>(shrunk);
  }
  return adopt(env, m.p, n, m.cap, false);
}

// A synchronous deflate() that finds no pooled block (a tight loop: Node runs no finalizer inside it) would hand the
// library ~64 MiB of untouched malloc'd memory to download into piece by piece — every page faulted under the copy
// engine's hands, ~40 ms.  Instead the library writes into ONE page-locked scratch block the addon keeps for this purpose,
// and the exact-length result is copied out of it by a few threads (the faults of the fresh result spread over them).
struct SyncScratch {
  uint8_t* p = nullptr;
  size_t cap = 0;
  uint8_t* get(size_t n) {
    if (n <= cap) return p;
    if (p) zes_host_free(p);
    p = nullptr;
    cap = 0;
    void* q = nullptr;
    const size_t want = (n + (4u << 20) - 1) & ~(size_t)((4u << 20) - 1);
    if (zes_host_alloc(want, &q) != 0 || !q) return nullptr;
    p = static_cast<uint8_t*>(q);
    cap = want;
    return p;
  }
  void clear() {
    if (p) zes_host_free(p);
    p = nullptr;
    cap = 0;
  }
};
SyncScratch g_sync_scratch;  // (JS thread only)
void copy_parallel(uint8_t* dst, const uint8_t* src, size_t n) {
  const unsigned hw = std::thread::hardware_concurrency();
  const unsigned nt = n < (8u << 20) ? 1u : (hw >= 8 ? 4u : 2u);
  if (nt == 1) {
    memcpy(dst, src, n);
    return;
  }
  std::vector<std::thread> th;
  const size_t part = ((n / nt) + 4095) & ~(size_t)4095;
  for (unsigned i = 1; i < nt; i++) {
    const size_t o = i * part;
    if (o < n) th.emplace_back([=] { memcpy(dst + o, src + o, (o + part < n) ? part : n - o); });
  }
  memcpy(dst, src, part < n ? part : n);
  for (auto& t : th) t.join();
}

// zes_alloc_fn of the synchronous inflate: runs on the JS thread, inside zes_inflate_alloc
struct SyncAlloc {
  ResultMem m;
  bool failed;
  bool in_scratch = false;  // the early estimate was served from the scratch block: the result is copied out of it afterwards
};
// an early request (ZES_F_ALLOC_BOUND: an upper estimate, the bytes then come down beside the decode) is only worth
// taking with a pooled block: downloads piece by piece into fresh malloc'd memory pay its page faults the slow way
ResultMem early_alloc(size_t n) {
  ResultMem m;
  if (n >= BIG_MIN) m.p = g_big.take(n, &m.cap, false);
  if (!m.p) m.cap = 0;
  return m;
}
uint8_t* sync_alloc(void* user, uint32_t index, uint64_t n) {
  SyncAlloc* a = static_cast<SyncAlloc*>(user);
  result_free(a->m);  // (ZES_F_ALLOC_BOUND: a second call when the first one's estimate fell short, or the call started over)
  a->m = ResultMem();
  a->in_scratch = false;
  if (index & ZES_ALLOC_EARLY) {
    // a pooled block if the pool holds one; else the scratch block when it is long enough (grown outside the library's call,
    // from what the calls before needed): the bytes come down beside the decode into page-locked memory, and the exact-length
    // result is copied out by a few threads afterwards — a tight loop's results are fresh memory either way, but this way
    // its page faults are spread over the copy's threads instead of being taken under the copy engine, and no block is
    // page-locked for the next call (a loop over 64 MiB results alternated between 2.5 ms and 25 ms a call)
    if (n >= BIG_MIN) a->m.p = g_big.take((size_t)n, &a->m.cap, false, false);
    if (a->m.p) return a->m.p;
    a->m.cap = 0;
    if (g_sync_scratch.p && n <= g_sync_scratch.cap) {
      a->in_scratch = true;
      return g_sync_scratch.p;
    }
    if (n >= BIG_MIN) g_big.take((size_t)n, &a->m.cap, false, true);  // (notes the miss: the pool is topped up after the call)
    a->m = ResultMem();
    return nullptr;  // ("not now" — the exact size is asked for once, later)
  }
  a->m = result_alloc((size_t)n, false);  // (inside the library's call)
  if (!a->m.p) a->failed = true;
  return a->m.p;
}

napi_value Deflate(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  const uint8_t* in = nullptr;
  size_t n = 0;
  if (argc < 1 || !get_bytes(env, argv[0], &in, &n)) {
    napi_throw_type_error(env, nullptr, "deflate(input): input must be a Uint8Array");
    return nullptr;
  }
  uint64_t cap = 0, out_len = 0;
  zes_deflate_bound(n, &cap);
  ResultMem tmp;
  if (cap >= BIG_MIN) tmp.p = g_big.take(cap, &tmp.cap, false);  // a block the pool holds, if any
  if (!tmp.p && cap >= (4u << 20)) {  // none (a tight synchronous loop): through the scratch block, see above
    tmp.cap = 0;
    uint8_t* sc = g_sync_scratch.get(cap);
    if (sc) {
      const int rc = zes_deflate(in, n, sc, cap, &out_len);
      if (rc) return throw_status(env, rc);
      ResultMem m;
      m.p = static_cast<uint8_t*>(malloc(out_len ? out_len : 1));
      if (!m.p) return throw_status(env, ZES_E_ARG);
      copy_parallel(m.p, sc, (size_t)out_len);
      return take_u8(env, m, out_len);
    }
  }
  if (!tmp.p) tmp = result_alloc(cap, false);
  if (!tmp.p) return throw_status(env, ZES_E_ARG);
  const int rc = zes_deflate(in, n, tmp.p, cap, &out_len);
  if (rc) {
    result_free(tmp);
    return throw_status(env, rc);
  }
  return take_u8(env, tmp, out_len);
}

napi_value Inflate(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  const uint8_t* in = nullptr;
  size_t c = 0;
  if (argc < 1 || !get_bytes(env, argv[0], &in, &c)) {
    napi_throw_type_error(env, nullptr, "inflate(input): input must be a Uint8Array");
    return nullptr;
  }
  // one call, one lock: the library decodes, then asks for the exact ArrayBuffer (the reference grows a
  // Uint8WriteStream instead, src/inflate.ts:17) and copies straight into it — nothing is kept between calls,
  // so an inflateAsync() in flight on a worker thread cannot get in between
  SyncAlloc sa;
  sa.failed = false;
  uint64_t out_len = 0;
  // the scratch block, long enough for what the last synchronous inflate produced (a loop decodes like after like) and
  // never shorter than the stream: grown here, outside the library's call
  static size_t s_last_out = 0;
  {
    const size_t want = s_last_out > c ? s_last_out + s_last_out / 16 + (1u << 20) : 0;
    if (want >= (4u << 20) && want <= (1024u << 20)) (void)g_sync_scratch.get(want);
  }
  // (the pooled block is longer than the ArrayBuffer shows of it anyway: the library may ask early for an upper estimate and
  // send the bytes down while it is still decoding)
  const int rc = zes_inflate_alloc(in, c, sync_alloc, &sa, &out_len, ZES_F_ALLOC_BOUND);
  if (!sa.in_scratch) g_big.top_up();
  if (rc) {
    if (!sa.in_scratch) result_free(sa.m);
    return throw_status(env, sa.failed ? ZES_E_ARG : rc);
  }
  s_last_out = (size_t)out_len;
  if (sa.in_scratch) {  // (the estimate held: the result is the scratch block's head)
    ResultMem m;
    if (out_len >= BIG_MIN) m.p = g_big.take((size_t)out_len, &m.cap, false, false);
    if (!m.p) {
      m.cap = 0;
      m.p = static_cast<uint8_t*>(malloc(out_len ? (size_t)out_len : 1));
    }
    if (!m.p) return throw_status(env, ZES_E_ARG);
    copy_parallel(m.p, g_sync_scratch.p, (size_t)out_len);
    return take_u8(env, m, out_len);
  }
  return take_u8(env, sa.m, out_len);
}

// deflateRaw(input): the raw stream of the reference's src/deflate.ts:14 (no zlib wrapper)
napi_value DeflateRaw(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  const uint8_t* in = nullptr;
  size_t n = 0;
  if (argc < 1 || !get_bytes(env, argv[0], &in, &n)) {
    napi_throw_type_error(env, nullptr, "deflateRaw(input): input must be a Uint8Array");
    return nullptr;
  }
  uint64_t cap = 0, out_len = 0;
  zes_deflate_bound(n, &cap);
  ResultMem tmp = result_alloc(cap);
  if (!tmp.p) return throw_status(env, ZES_E_ARG);
  const int rc = zes_deflate_raw(in, n, tmp.p, cap, &out_len);
  if (rc) {
    result_free(tmp);
    return throw_status(env, rc);
  }
  return take_u8(env, tmp, out_len);
}

// inflateRaw(input, offset = 0): the reference's src/inflate.ts:16 (what src/zlib.ts:21 calls with offset 2)
napi_value InflateRaw(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  const uint8_t* in = nullptr;
  size_t c = 0;
  if (argc < 1 || !get_bytes(env, argv[0], &in, &c)) {
    napi_throw_type_error(env, nullptr, "inflateRaw(input, offset): input must be a Uint8Array");
    return nullptr;
  }
  uint64_t offset = 0;
  if (argc >= 2) {
    napi_valuetype vt;
    double d = 0;
    if (napi_typeof(env, argv[1], &vt) != napi_ok || (vt != napi_number && vt != napi_undefined) ||
        (vt == napi_number && (napi_get_value_double(env, argv[1], &d) != napi_ok || !(d >= 0) || d > 9007199254740991.0 || d != (double)(uint64_t)d))) {
      napi_throw_type_error(env, nullptr, "inflateRaw(input, offset): offset must be a non-negative safe integer");
      return nullptr;
    }
    offset = (uint64_t)d;
  }
  // grow-and-retry like the reference's Uint8WriteStream (src/utils/Uint8WriteStream.ts:13-21)
  uint64_t cap = c * 4 + 65536, out_len = 0;
  for (int attempt = 0; attempt < 8; attempt++) {
    ResultMem tmp = result_alloc(cap);
    if (!tmp.p) return throw_status(env, ZES_E_ARG);
    const int rc = zes_inflate_raw(in, c, offset, tmp.p, cap, &out_len, ZES_F_DEFAULT);
    if (rc == ZES_E_NOSPACE && out_len > cap) {
      result_free(tmp);
      cap = out_len;
      continue;
    }
    if (rc) {
      result_free(tmp);
      return throw_status(env, rc);
    }
    return take_u8(env, tmp, out_len);
  }
  return throw_status(env, ZES_E_DEVICE);
}

napi_value Adler32(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  const uint8_t* in = nullptr;
  size_t n = 0;
  if (argc < 1 || !get_bytes(env, argv[0], &in, &n)) {
    napi_throw_type_error(env, nullptr, "adler32(input): input must be a Uint8Array");
    return nullptr;
  }
  uint32_t a = 0;
  const int rc = zes_adler32(in, n, &a);
  if (rc) return throw_status(env, rc);
  napi_value v;
  napi_create_uint32(env, a, &v);
  return v;
}

napi_value Init(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  int32_t dev = 0;
  if (argc >= 1) napi_get_value_int32(env, argv[0], &dev);
  const int rc = zes_init(dev);
  if (rc) return throw_status(env, rc);
  napi_value v;
  napi_get_undefined(env, &v);
  return v;
}

// initDevices(n): one process, n GPUs (n omitted or <= 0: every visible one) — the batch calls then partition their
// buffers over all of them, single calls take them in turn (zes_init_devices).  Returns the number of devices in use.
napi_value InitDevices(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  int32_t n = 0;
  if (argc >= 1) napi_get_value_int32(env, argv[0], &n);
  const int rc = zes_init_devices(n);
  if (rc) return throw_status(env, rc);
  napi_value v;
  napi_create_int32(env, zes_device_count(), &v);
  return v;
}

// trim(): the library's pooled device scratch goes back to the driver (zes_trim); the next call allocates again
napi_value Trim(napi_env env, napi_callback_info) {
  g_big.clear();
  g_sync_scratch.clear();
  const int rc = zes_trim();
  if (rc) return throw_status(env, rc);
  napi_value v;
  napi_get_undefined(env, &v);
  return v;
}

// ---- Promise-returning forms: the blocking C-ABI call runs on a libuv worker thread ----
struct AsyncJob {
  napi_async_work work = nullptr;
  napi_deferred deferred = nullptr;
  napi_ref input_ref = nullptr;  // keeps the caller's array alive (and its memory in place) while the worker reads it
  const uint8_t* in = nullptr;
  size_t n = 0;
  bool inflate = false;
  ResultMem out;
  uint64_t out_len = 0;
  int rc = 0;
};

uint8_t* async_alloc(void* user, uint32_t index, uint64_t n) {  // zes_alloc_fn on the worker thread: pool or malloc, no N-API calls
  AsyncJob* j = static_cast<AsyncJob*>(user);
  result_free(j->out);  // (ZES_F_ALLOC_BOUND: see sync_alloc)
  if (index & ZES_ALLOC_EARLY) {
    j->out = early_alloc((size_t)n);
    return j->out.p;
  }
  j->out = result_alloc((size_t)n, false);  // (inside the library's call)
  return j->out.p;
}

void async_execute(napi_env, void* data) {  // worker thread: no N-API calls here
  AsyncJob* j = static_cast<AsyncJob*>(data);
  if (!j->inflate) {
    uint64_t cap = 0;
    zes_deflate_bound(j->n, &cap);
    j->out = result_alloc(cap ? cap : 1);
    j->rc = j->out.p ? zes_deflate(j->in, j->n, j->out.p, cap, &j->out_len) : ZES_E_ARG;
    return;
  }
  // one call: the library decodes, then asks for memory of the exact size (no state is kept in the library between calls)
  j->rc = zes_inflate_alloc(j->in, j->n, async_alloc, j, &j->out_len, ZES_F_ALLOC_BOUND);
  g_big.top_up();
}


void async_complete(napi_env env, napi_status, void* data) {  // JS thread again
  AsyncJob* j = static_cast<AsyncJob*>(data);
  napi_value result = nullptr;
  bool ok = j->rc == 0;
  if (ok) {
    // the worker's memory becomes the result's ArrayBuffer (exact length, no copy on the JS thread)
    result = take_u8(env, j->out, (size_t)j->out_len);
    j->out = ResultMem();  // owned by the ArrayBuffer now (or given back by take_u8)
    if (!result) {
      ok = false;
      j->rc = ZES_E_ARG;
    }
  }
  if (ok) {
    napi_resolve_deferred(env, j->deferred, result);
  } else {
    napi_value msg, err;
    napi_create_string_utf8(env, zes_strerror(j->rc), NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, nullptr, msg, &err);  // plain Error with the reference's message, as a rejection
    napi_reject_deferred(env, j->deferred, err);
  }
  result_free(j->out);
  napi_delete_reference(env, j->input_ref);
  napi_delete_async_work(env, j->work);
  delete j;
}

napi_value start_async(napi_env env, napi_callback_info info, bool inflate) {
  size_t argc = 1;
  napi_value argv[1];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  const uint8_t* in = nullptr;
  size_t n = 0;
  if (argc < 1 || !get_bytes(env, argv[0], &in, &n)) {
    napi_throw_type_error(env, nullptr, inflate ? "inflateAsync(input): input must be a Uint8Array" : "deflateAsync(input): input must be a Uint8Array");
    return nullptr;
  }
  AsyncJob* j = new AsyncJob();
  j->in = in;
  j->n = n;
  j->inflate = inflate;
  napi_value promise, name;
  if (napi_create_promise(env, &j->deferred, &promise) != napi_ok) {
    delete j;
    napi_throw_error(env, nullptr, "zes: could not create a promise");
    return nullptr;
  }
  // The reference on the array keeps it (and its memory) alive while the worker reads it.  The caller must not
  // transfer or detach its ArrayBuffer before the promise settles (documented in zlib.ts): N-API v3 has no way to
  // pin a backing store against a transfer.
  bool ok = napi_create_reference(env, argv[0], 1, &j->input_ref) == napi_ok &&
            napi_create_string_utf8(env, inflate ? "zes_inflate" : "zes_deflate", NAPI_AUTO_LENGTH, &name) == napi_ok &&
            napi_create_async_work(env, nullptr, name, async_execute, async_complete, j, &j->work) == napi_ok;
  if (ok && napi_queue_async_work(env, j->work) != napi_ok) {
    napi_delete_async_work(env, j->work);
    ok = false;
  }
  if (!ok) {  // nothing was queued: settle the promise here, free the job
    napi_value msg, err;
    napi_create_string_utf8(env, "zes: could not queue the work", NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, nullptr, msg, &err);
    napi_reject_deferred(env, j->deferred, err);
    if (j->input_ref) napi_delete_reference(env, j->input_ref);
    delete j;
  }
  return promise;
}
napi_value DeflateAsync(napi_env env, napi_callback_info info) { return start_async(env, info, false); }
napi_value InflateAsync(napi_env env, napi_callback_info info) { return start_async(env, info, true); }

// ---- batch forms (deflateBatch / inflateBatch): an array of independent buffers in one call, so that
// small buffers share the launches and fill the chip together (zes_deflate_batch / zes_inflate_batch_alloc) ----
struct BatchJob {
  napi_async_work work = nullptr;
  napi_deferred deferred = nullptr;  // null: synchronous call
  napi_ref input_ref = nullptr;
  bool inflate = false;
  uint32_t count = 0;
  const uint8_t** in = nullptr;
  uint64_t* in_len = nullptr;
  uint8_t** out = nullptr;
  size_t* out_pool = nullptr;  // capacity of the pooled block behind out[i], or 0: malloc'd
  uint64_t* out_cap = nullptr;
  uint64_t* out_len = nullptr;
  int32_t* status = nullptr;
  int rc = 0;
  ~BatchJob() {
    if (out)
      for (uint32_t i = 0; i < count; i++) {
        ResultMem m;
        m.p = out[i];
        m.cap = out_pool ? out_pool[i] : 0;
        result_free(m);
      }
    delete[] out_pool;
    delete[] in;
    delete[] in_len;
    delete[] out;
    delete[] out_cap;
    delete[] out_len;
    delete[] status;
  }
};

uint8_t* batch_alloc(void* user, uint32_t i, uint64_t n) {  // any thread: plain malloc, wrapped into an ArrayBuffer later
  BatchJob* j = static_cast<BatchJob*>(user);
  const ResultMem m = result_alloc((size_t)n, false);  // (inside the library's call, possibly on one of its threads)
  j->out[i] = m.p;
  j->out_pool[i] = m.cap;
  return j->out[i];
}

void batch_execute(napi_env, void* data) {
  BatchJob* j = static_cast<BatchJob*>(data);
  if (j->inflate) {
    j->rc = zes_inflate_batch_alloc(j->in, j->in_len, batch_alloc, j, j->out_len, j->status, j->count, ZES_F_DEFAULT);
    return;
  }
  for (uint32_t i = 0; i < j->count; i++) {
    zes_deflate_bound(j->in_len[i], &j->out_cap[i]);
    const ResultMem m = result_alloc((size_t)j->out_cap[i]);
    j->out[i] = m.p;
    j->out_pool[i] = m.cap;
    if (!j->out[i]) {
      j->rc = ZES_E_ARG;
      return;
    }
  }
  j->rc = zes_deflate_batch(j->in, j->in_len, j->out, j->out_cap, j->out_len, j->status, j->count);
}

// results: an array with, per buffer, a fresh Uint8Array or an Error carrying the reference's message
// (a batch never throws for one bad buffer: the caller sees which ones failed)
napi_value batch_results(napi_env env, BatchJob* j) {
  napi_value arr;
  if (napi_create_array_with_length(env, j->count, &arr) != napi_ok) return nullptr;
  for (uint32_t i = 0; i < j->count; i++) {
    napi_value v = nullptr;
    if (j->status[i] == 0) {
      ResultMem m;
      m.p = j->out[i];
      m.cap = j->out_pool[i];
      j->out[i] = nullptr;  // owned by the ArrayBuffer from here on (or given back by take_u8)
      j->out_pool[i] = 0;
      v = take_u8(env, m, (size_t)j->out_len[i]);
      if (!v) return nullptr;
    } else {
      napi_value msg;
      napi_create_string_utf8(env, zes_strerror(j->status[i]), NAPI_AUTO_LENGTH, &msg);
      napi_create_error(env, nullptr, msg, &v);
    }
    napi_set_element(env, arr, i, v);
  }
  return arr;
}

void batch_complete(napi_env env, napi_status, void* data) {
  BatchJob* j = static_cast<BatchJob*>(data);
  napi_value res = j->rc == 0 ? batch_results(env, j) : nullptr;
  if (res) {
    napi_resolve_deferred(env, j->deferred, res);
  } else {
    napi_value msg, err;
    napi_create_string_utf8(env, zes_strerror(j->rc ? j->rc : ZES_E_ARG), NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, nullptr, msg, &err);
    napi_reject_deferred(env, j->deferred, err);
  }
  napi_delete_reference(env, j->input_ref);
  napi_delete_async_work(env, j->work);
  delete j;
}

napi_value start_batch(napi_env env, napi_callback_info info, bool inflate, bool async) {
  size_t argc = 1;
  napi_value argv[1];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  bool is_arr = false;
  uint32_t count = 0;
  if (argc < 1 || napi_is_array(env, argv[0], &is_arr) != napi_ok || !is_arr || napi_get_array_length(env, argv[0], &count) != napi_ok) {
    napi_throw_type_error(env, nullptr, "batch(inputs): inputs must be an array of Uint8Array");
    return nullptr;
  }
  BatchJob* j = new BatchJob();
  j->inflate = inflate;
  j->count = count;
  j->in = new const uint8_t*[count + 1]();
  j->in_len = new uint64_t[count + 1]();
  j->out = new uint8_t*[count + 1]();
  j->out_pool = new size_t[count + 1]();
  j->out_cap = new uint64_t[count + 1]();
  j->out_len = new uint64_t[count + 1]();
  j->status = new int32_t[count + 1]();
  for (uint32_t i = 0; i < count; i++) {
    napi_value e;
    size_t n = 0;
    if (napi_get_element(env, argv[0], i, &e) != napi_ok || !get_bytes(env, e, &j->in[i], &n)) {
      delete j;
      napi_throw_type_error(env, nullptr, "batch(inputs): every element must be a Uint8Array");
      return nullptr;
    }
    j->in_len[i] = n;
  }
  if (!async) {
    batch_execute(env, j);
    napi_value res = j->rc == 0 ? batch_results(env, j) : throw_status(env, j->rc);
    delete j;
    return res;
  }
  napi_value promise, name;
  if (napi_create_promise(env, &j->deferred, &promise) != napi_ok) {
    delete j;
    napi_throw_error(env, nullptr, "zes: could not create a promise");
    return nullptr;
  }
  bool ok = napi_create_reference(env, argv[0], 1, &j->input_ref) == napi_ok &&  // the outer array keeps every element alive
            napi_create_string_utf8(env, inflate ? "zes_inflate_batch" : "zes_deflate_batch", NAPI_AUTO_LENGTH, &name) == napi_ok &&
            napi_create_async_work(env, nullptr, name, batch_execute, batch_complete, j, &j->work) == napi_ok;
  if (ok && napi_queue_async_work(env, j->work) != napi_ok) {
    napi_delete_async_work(env, j->work);
    ok = false;
  }
  if (!ok) {
    napi_value msg, err;
    napi_create_string_utf8(env, "zes: could not queue the work", NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, nullptr, msg, &err);
    napi_reject_deferred(env, j->deferred, err);
    if (j->input_ref) napi_delete_reference(env, j->input_ref);
    delete j;
  }
  return promise;
}
napi_value DeflateBatch(napi_env env, napi_callback_info info) { return start_batch(env, info, false, false); }
napi_value InflateBatch(napi_env env, napi_callback_info info) { return start_batch(env, info, true, false); }
napi_value DeflateBatchAsync(napi_env env, napi_callback_info info) { return start_batch(env, info, false, true); }
napi_value InflateBatchAsync(napi_env env, napi_callback_info info) { return start_batch(env, info, true, true); }

// allocPinned(n): a Uint8Array in page-locked memory (zes_host_alloc) — buffers from here cross PCIe without the
// library's staging copy; released when the array is collected.  The finalizer runs on the JS thread and takes
// the library's lock for a moment (no stream is waited for: hipHostFree itself waits for work that uses the block), so
// it can stall behind a call that is holding the lock; zes_host_free also works after zes_shutdown.
napi_value AllocPinned(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  double d = -1;
  if (argc < 1 || napi_get_value_double(env, argv[0], &d) != napi_ok || !(d >= 0) || d > 1e12 || d != (double)(uint64_t)d) {
    napi_throw_type_error(env, nullptr, "allocPinned(n): n must be a non-negative integer");
    return nullptr;
  }
  void* p = nullptr;
  const int rc = zes_host_alloc((uint64_t)d, &p);
  if (rc) return throw_status(env, rc);
  return adopt(env, static_cast<uint8_t*>(p), (size_t)d, 0, true);  // (released by a later call's sweep once the array is collected)
}

// every entry point first looks which arrays the collector has taken since the last call
template <napi_callback F>
napi_value swept(napi_env env, napi_callback_info info) {
  sweep(env);
  return F(env, info);
}

napi_value ModuleInit(napi_env env, napi_value exports) {
  napi_property_descriptor props[] = {
      {"deflate", nullptr, swept<Deflate>, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"inflate", nullptr, swept<Inflate>, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"deflateRaw", nullptr, swept<DeflateRaw>, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"inflateRaw", nullptr, swept<InflateRaw>, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"deflateAsync", nullptr, swept<DeflateAsync>, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"inflateAsync", nullptr, swept<InflateAsync>, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"deflateBatch", nullptr, swept<DeflateBatch>, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"inflateBatch", nullptr, swept<InflateBatch>, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"deflateBatchAsync", nullptr, swept<DeflateBatchAsync>, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"inflateBatchAsync", nullptr, swept<InflateBatchAsync>, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"allocPinned", nullptr, swept<AllocPinned>, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"adler32", nullptr, swept<Adler32>, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"init", nullptr, swept<Init>, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"initDevices", nullptr, swept<InitDevices>, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"trim", nullptr, swept<Trim>, nullptr, nullptr, nullptr, napi_default, nullptr},
  };
  napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props);
  if (getenv("ZES_NAPI_SEGV_TRACE")) {  // development: where a SIGSEGV comes from
    signal(SIGSEGV, [](int) {
      void* bt[64];
      const int n = backtrace(bt, 64);
      backtrace_symbols_fd(bt, n, 2);
      _exit(139);
    });
  }
  g_envs.fetch_add(1);
  g_exiting.store(false);
  {
    EnvTracks* et = new EnvTracks;
    et->env = env;
    std::lock_guard<std::mutex> lk(g_tracks_mu);
    g_tracks.push_back(et);
  }
  napi_add_env_cleanup_hook(env, env_cleanup, env);
  static bool once = false;
  if (!once) {
    once = true;
    atexit(mark_exiting_atexit);
  }
  return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, ModuleInit)
