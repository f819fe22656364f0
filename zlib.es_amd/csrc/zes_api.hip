// zes_api.hip — host side of the C-ABI declared in include/zes.h.
//
// One process drives one GPU (zes_init(device)); all launches go to one private HIP stream.
// Device scratch is pooled and only grows.  No CPU fallback exists here: every compute entry
// point needs a working gfx950 device and returns ZES_E_DEVICE otherwise.
#include "../../include/zes.h"
#include "zes_kernels.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <functional>
#include <future>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

extern "C" int zes_gen(uint8_t* out, uint64_t n, uint32_t kind, uint32_t seed);

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct KTime {
  double ms = 0;
  uint32_t launches = 0;
};

struct Ctx {
  bool ready = false;
  int device = -1;
  int want = -1;  // zes_init_devices: the device this context binds to at its first use
  hipStream_t stream = nullptr;
  hipStream_t cs_in = nullptr, cs_out = nullptr;  // copy streams of the pipelined host calls (H2D / D2H beside the kernels)
  hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_k = nullptr;
  hipEvent_t ev_rng[2] = {nullptr, nullptr};  // host inflate in pieces: the results of piece k have come back
  hipStream_t s_adler = nullptr;                     // the Adler-32 pass of a deflate call runs beside the LZ77 kernels
  hipEvent_t ev_a0 = nullptr, ev_a1 = nullptr;
  // deflate scratch
  DevBuf bufs, blks, idx_a, idx_b, sdelta, tmask, mlist, hists, codes, hdrs, adler, res, order;
  // inflate scratch
  DevBuf surv, vlong, segfail, symoff, cand, cand_sorted, counters, cres, map, resume, dbg, ibufs, ibufs2, mvlist, scratch, sres, maps, seglist, segprefix, wins, sym16, segorder, segjobs, pw16, gwins, seglive, segouts;
  // staging for the host-pointer API
  DevBuf st_in, st_out;
  DevBuf kraft;  // k_inf_scan's table: Kraft contribution of four 3-bit code-length fields at once
  void* pinned = nullptr;  // small pinned area for read-backs
  size_t pinned_cap = 0;
  void* mirror = nullptr;  // one-buffer inflate: the block decoder's results as the host reads them (ZesParMirror)
  // profiling
  bool profiling = false;
  std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending;
  std::vector<hipEvent_t> event_pool;
  std::vector<std::pair<std::string, KTime>> last_times;
  std::vector<std::string> name_pool;
  int last_tier = 0;
  // the survivor list of the block-start search as the block-parallel tier's scan left it (one buffer): the segment-parallel
  // tier's search of the same stream starts from it instead of scanning again
  bool sv_ok = false;
  const uint8_t* sv_din = nullptr;
  const void* sv_list = nullptr;  // (g.surv.p when the list was made: a pool that has grown since holds something else)
  uint64_t sv_in_off = 0, sv_c = 0;
  uint32_t sv_n = 0;
  char arch[64] = {0};
  int cus = 0;
  uint64_t hbm = 0;
};

// One context per device the library drives (zes_init(ngpus); "the batch API is where multi-GPU
// concurrency lives").  zes_init(device) binds context 0 — one process per GPU, what bench.py's ranks do;
// zes_init_devices(n) binds contexts 0 .. n-1 to devices 0 .. n-1, and the host-pointer entry points spread their
// work over them: a batch is partitioned by size (zes_partition, the rule of shard.partition) and every share runs
// on its own host thread against its own context — stream, scratch pools, staging ring, lock — while single calls
// go round robin.  Results of the host forms land in the caller's memory, so no device-to-device gather is needed
// here; HBM-resident results are gathered by shard.py over RCCL.  Which context a thread works on is thread-local.
constexpr int ZES_MAX_DEV = 16;
Ctx g_ctx[ZES_MAX_DEV];
std::mutex g_mus[ZES_MAX_DEV];
std::atomic<int> g_nctx{1};     // contexts in use (written under g_cfg_mu; read by every routed call)
std::mutex g_cfg_mu;             // guards g_nctx and the context -> device binding
thread_local int t_dev = 0;      // the context of the call this thread is inside
thread_local bool t_routed = false;
thread_local int t_last = 0;     // the context that served this thread's last call (what zes_last_inflate_tier / zes_last_kernel_times report on)
#define g (g_ctx[t_dev])
#define g_mu (g_mus[t_dev])
struct UseDev {                  // a call's context for its duration (nested entry points keep the outer one's)
  int prev;
  bool prev_routed;
  explicit UseDev(int d) : prev(t_dev), prev_routed(t_routed) {
    if (!t_routed) t_dev = d;
    t_routed = true;
    t_last = t_dev;
  }
  ~UseDev() {
    t_dev = prev;
    t_routed = prev_routed;
  }
};

#define HIPCHK(x)                                                                              \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      if (getenv("ZES_DEBUG")) fprintf(stderr, "zes: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
      return ZES_E_DEVICE;                                                                     \
    }                                                                                          \
  } while (0)

// pinned staging: [0, PIN_UP) read-back area, [PIN_UP, pinned_cap) upload area for the buffer table
constexpr size_t PIN_UP = 256 << 10;
constexpr size_t MIRROR_ITEMS = 8192;  // work items whose results the host mirror holds (a call with a larger launch bound keeps the chain kernel)

int ensure(DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return ZES_OK;
  if (b.p) HIPCHK(hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  size_t want = bytes + bytes / 8 + 4096;
  HIPCHK(hipMalloc(&b.p, want));
  b.cap = want;
  return ZES_OK;
}

int init_locked(int device) {
  if (g.ready) {
    if (device >= 0 && device != g.device) return ZES_E_ARG;
    // HIP's current device is per thread: a call from a thread that never made one current (a libuv worker, a Python
    // thread) would allocate on device 0 while the stream belongs to g.device
    HIPCHK(hipSetDevice(g.device));
    return ZES_OK;
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return ZES_E_DEVICE;
  if (device < 0) device = g.want >= 0 ? g.want : 0;
  if (device >= n) return ZES_E_DEVICE;
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  snprintf(g.arch, sizeof g.arch, "%s", prop.gcnArchName);
  g.cus = prop.multiProcessorCount;
  g.hbm = prop.totalGlobalMem;
  HIPCHK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&g.cs_in, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&g.cs_out, hipStreamNonBlocking));
  for (int k = 0; k < 2; k++) HIPCHK(hipEventCreateWithFlags(&g.ev_up[k], hipEventDisableTiming));
  for (int k = 0; k < 2; k++) HIPCHK(hipEventCreateWithFlags(&g.ev_rng[k], hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&g.ev_k, hipEventDisableTiming));
  HIPCHK(hipStreamCreateWithFlags(&g.s_adler, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&g.ev_a0, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&g.ev_a1, hipEventDisableTiming));
  g.pinned_cap = 1 << 20;
  HIPCHK(hipHostMalloc(&g.pinned, g.pinned_cap, hipHostMallocDefault));
  if (hipHostMalloc(&g.mirror, MIRROR_ITEMS * (sizeof(ZesCandRes) + 4), hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    g.mirror = nullptr;  // (the chain kernel does the work then)
  }
  {
    // units of 2^-7, a field of 0 adds nothing; saturated at 200 so that an over-full group can never sum back to exactly 128
    uint8_t tab[4096];
    for (uint32_t i = 0; i < 4096; i++) {
      uint32_t k = 0;
      for (uint32_t f = 0; f < 4; f++) k += (128u >> ((i >> (3 * f)) & 7u)) & 127u;
      tab[i] = (uint8_t)std::min(k, 200u);
    }
    HIPCHK(hipMalloc(&g.kraft.p, 4096));
    g.kraft.cap = 4096;
    HIPCHK(hipMemcpy(g.kraft.p, tab, 4096, hipMemcpyHostToDevice));
  }
  g.device = device;
  g.ready = true;
  return ZES_OK;
}

// ---- which context serves a call ----
int route_host() {  // host-pointer work: the contexts in turn
  if (t_routed) return t_dev;
  const int n = g_nctx.load();
  if (n <= 1) return 0;
  static std::atomic<uint32_t> rr{0};
  return (int)(rr.fetch_add(1) % (uint32_t)n);
}
// -1: several contexts are bound and none of them drives the device that holds the memory (or the pointer is not device
// memory at all, or the two pointers of a call sit on different devices) — the entry point answers ZES_E_ARG instead of
// letting context 0's kernels touch memory of a device it does not drive
int route_dev(const void* p, const void* p2 = nullptr) {  // device-pointer work: the context of the device that holds the memory
  if (t_routed) return t_dev;
  if (g_nctx.load() <= 1 || !p) return 0;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  if (p2) {
    hipPointerAttribute_t b;
    if (hipPointerGetAttributes(&b, p2) != hipSuccess) {
      (void)hipGetLastError();
      return -1;
    }
    if (b.device != a.device) return -1;
  }
  for (int i = 0, n = g_nctx.load(); i < n; i++)
    if ((g_ctx[i].ready ? g_ctx[i].device : g_ctx[i].want) == a.device) return i;
  return -1;
}
#define ROUTE_DEV(...)                    \
  const int rd_ = route_dev(__VA_ARGS__); \
  if (rd_ < 0) return ZES_E_ARG;          \
  UseDev ud(rd_)

// ---- host <-> device staging of the host-pointer entry points ----
// A caller's buffer (a JS Uint8Array, a numpy array) is pageable: the DMA engines cannot read it.  It crosses in
// chunks through a ring of pinned buffers: helper threads copy chunk k+1 into its pinned buffer while the DMA of
// chunk k runs (and the other way round on the way back), so the trip costs max(memcpy, DMA) instead of their sum
// and the memcpy is spread over several cores.  A buffer that already is pinned (zes_host_alloc) is handed to the
// DMA engine as it is.
constexpr size_t STAGE_CHUNK = 8u << 20;
constexpr int STAGE_RING = 4;
constexpr size_t STAGE_DIRECT_MAX = 256u << 10;  // below this one plain copy call is quicker than the ring

struct CopyPool {
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  uint8_t* dst = nullptr;
  const uint8_t* src = nullptr;
  size_t n = 0, part = 0;
  uint32_t gen = 0, parts = 0;
  std::atomic<uint32_t> next{0};
  uint32_t done = 0, active = 0;
  bool stop = false;

  void worker() {
    uint32_t seen = 0;
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv_work.wait(lk, [&] { return stop || gen != seen; });
      if (stop) return;
      seen = gen;
      active++;  // checked in under the lock: copy() does not return (and the next job is not written) before every
      lk.unlock();  // worker that picked this generation up has checked out again — no claim of an old job can
      const uint32_t did = run_parts();  // meet the fields of a new one
      lk.lock();
      done += did;
      active--;
      if (done >= parts && active == 0) cv_done.notify_all();
    }
  }
  uint32_t run_parts() {
    uint32_t did = 0;
    for (;;) {
      const uint32_t k = next.fetch_add(1);
      if (k >= parts) return did;
      const size_t o = (size_t)k * part;
      memcpy(dst + o, src + o, std::min(part, n - o));
      did++;
    }
  }
  void start() {
    if (!threads.empty()) return;
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned nt = std::max(1u, std::min(7u, hw ? hw / 2 : 2u));  // (two pools + two side threads: 16, a one-GPU box's share)
    for (unsigned i = 0; i + 1 < nt; i++) threads.emplace_back([this] { worker(); });
  }
  // memcpy spread over the pool (the caller takes parts too); plain memcpy when it is short
  void copy(uint8_t* d, const uint8_t* s_, size_t bytes) {
    if (bytes < (512u << 10) || threads.empty()) {
      memcpy(d, s_, bytes);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      dst = d;
      src = s_;
      n = bytes;
      part = 256u << 10;
      parts = (uint32_t)((bytes + part - 1) / part);
      next.store(0);
      done = 0;
      gen++;
    }
    cv_work.notify_all();
    const uint32_t did = run_parts();
    std::unique_lock<std::mutex> lk(mu);
    done += did;
    cv_done.wait(lk, [&] { return done >= parts && active == 0; });
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_work.notify_all();
    for (auto& t : threads) t.join();
    threads.clear();
    stop = false;
  }
  // a process that never calls zes_shutdown still has to get rid of the helpers: destroying a joinable std::thread
  // at exit calls std::terminate (seen as a host process that never exits)
  ~CopyPool() { shutdown(); }
};

// One direction's staging: a ring of pinned chunks, their events, the copy helpers.  Two of them, so that an upload
// and a download of a pipelined call run side by side (each on its own copy stream and its own thread).
struct Stager {
  CopyPool pool;
  uint8_t* buf[STAGE_RING] = {nullptr};
  hipEvent_t ev[STAGE_RING] = {nullptr};
  int ready() {
    if (buf[0]) return ZES_OK;
    for (int k = 0; k < STAGE_RING; k++) {
      HIPCHK(hipHostMalloc((void**)&buf[k], STAGE_CHUNK, hipHostMallocDefault));
      HIPCHK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    }
    pool.start();
    return ZES_OK;
  }
  void release() {
    pool.shutdown();
    for (int k = 0; k < STAGE_RING; k++) {
      if (buf[k]) (void)hipHostFree(buf[k]);
      if (ev[k]) (void)hipEventDestroy(ev[k]);
      buf[k] = nullptr;
      ev[k] = nullptr;
    }
  }
};
Stager g_ups[ZES_MAX_DEV], g_downs[ZES_MAX_DEV];
#define g_up (g_ups[t_dev])
#define g_down (g_downs[t_dev])

// A thread that runs the side legs of a pipelined host call (one for uploads, one for downloads): submit() hands it
// a task, the returned future gives the task's status.
struct SideThread {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::packaged_task<int()>> q;
  bool stop = false;
  int owner = 0;
  void loop() {
    t_dev = owner;                                     // the context this thread serves
    if (g.device >= 0) (void)hipSetDevice(g.device);  // HIP's current device is per thread
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return stop || !q.empty(); });
      if (q.empty()) return;
      std::packaged_task<int()> t = std::move(q.front());
      q.erase(q.begin());
      lk.unlock();
      t();
      lk.lock();
    }
  }
  std::future<int> submit(std::function<int()> fn) {
    std::packaged_task<int()> t(std::move(fn));
    std::future<int> f = t.get_future();
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!th.joinable()) {
        owner = t_dev;
        th = std::thread([this] { loop(); });
      }
      q.push_back(std::move(t));
    }
    cv.notify_one();
    return f;
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
    stop = false;
  }
  ~SideThread() { shutdown(); }
};
SideThread g_side_ups[ZES_MAX_DEV], g_side_downs[ZES_MAX_DEV];
#define g_side_up (g_side_ups[t_dev])
#define g_side_down (g_side_downs[t_dev])

bool is_pinned(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // an ordinary malloc'd pointer: not an error of ours
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

// A caller's large pageable buffer goes through the runtime's own copy call: on the MI355X host hipMemcpy reads and
// writes pageable memory at the rate of pinned memory (1.2 ms per 64 MiB either way, tools/gpu_hostregister_probe.py:
// it page-locks the range for the copy itself), where the staging ring below costs a memcpy of every byte (~1 ms per
// 32 MiB with seven helper threads).  The copy is complete when upload()/download() return; in the pipelined calls
// they run on the side threads.  (Page-locking the caller's buffer in place with hipHostRegister for the whole call
// was as fast — 0.25 ms per 64 MiB to register — and was taken out when a GPU memory fault turned up in a long fuzz
// run with it.  That fault was traced later (ZES_TRACE_KERNELS tail of tools/gpu_fuzz.py seed 77) to k_inf_verify
// reading stale surv[] entries the scan had reserved but not written — fixed in the scan, DESIGN §6 — so registration
// was not its cause.  It stays off because it buys nothing over the runtime's path on this host and would pin a
// caller's pages for the length of the call.)
// The ring remains for buffers of 256 KiB to 4 MiB: the per-call cost of the runtime's path shows there.
constexpr uint64_t RUNTIME_COPY_MIN = 4ull << 20;

// host -> device on `stream` (the library's stream by default).  A pageable source has been read when this returns;
// a PINNED source (or one of <= STAGE_DIRECT_MAX bytes, which the runtime stages itself) is only enqueued: the DMA
// reads it asynchronously, and every caller synchronises `stream` before it returns to its own caller (they all do:
// each entry point ends with the read-back of its result on g.stream, the pipelined ones join their side threads).
int upload(uint8_t* d_dst, const uint8_t* src, uint64_t n, hipStream_t stream = nullptr) {
  if (!stream) stream = g.stream;
  if (!n) return ZES_OK;
  if (n <= STAGE_DIRECT_MAX || is_pinned(src)) {
    HIPCHK(hipMemcpyAsync(d_dst, src, n, hipMemcpyHostToDevice, stream));
    return ZES_OK;
  }
  if (n >= RUNTIME_COPY_MIN && !getenv("ZES_STAGE_RING")) {
    HIPCHK(hipMemcpyAsync(d_dst, src, n, hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));  // the caller's memory has been read
    return ZES_OK;
  }
  Stager& S = g_up;
  int rc = S.ready();
  if (rc) return rc;
  uint64_t off = 0;
  for (uint32_t k = 0; off < n; k++, off += STAGE_CHUNK) {
    const int slot = (int)(k % STAGE_RING);
    const size_t len = (size_t)std::min<uint64_t>(STAGE_CHUNK, n - off);
    if (k >= STAGE_RING) HIPCHK(hipEventSynchronize(S.ev[slot]));  // its previous DMA has read the pinned buffer
    S.pool.copy(S.buf[slot], src + off, len);
    HIPCHK(hipMemcpyAsync(d_dst + off, S.buf[slot], len, hipMemcpyHostToDevice, stream));
    HIPCHK(hipEventRecord(S.ev[slot], stream));
  }
  // the pinned ring is reused by the next call: its DMAs must have left it (the caller's memory was read above)
  for (int k = 0; k < STAGE_RING; k++) HIPCHK(hipEventSynchronize(S.ev[k]));
  return ZES_OK;
}

// device -> host, complete on return
// (`wait`: with a pinned destination, return with the copy in flight on `stream`; the caller synchronises)
int download(uint8_t* dst, const uint8_t* d_src, uint64_t n, hipStream_t stream = nullptr, bool wait = true) {
  if (!stream) stream = g.stream;
  const bool direct = n <= STAGE_DIRECT_MAX || is_pinned(dst);
  if (n && direct) HIPCHK(hipMemcpyAsync(dst, d_src, n, hipMemcpyDeviceToHost, stream));
  if (direct) {
    if (wait) HIPCHK(hipStreamSynchronize(stream));
    return ZES_OK;
  }
  if (n >= RUNTIME_COPY_MIN && !getenv("ZES_STAGE_RING")) {
    HIPCHK(hipMemcpyAsync(dst, d_src, n, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return ZES_OK;
  }
  Stager& S = g_down;
  int rc = S.ready();
  if (rc) return rc;
  const uint32_t chunks = (uint32_t)((n + STAGE_CHUNK - 1) / STAGE_CHUNK);
  auto issue = [&](uint32_t k) -> int {
    const uint64_t off = (uint64_t)k * STAGE_CHUNK;
    const size_t len = (size_t)std::min<uint64_t>(STAGE_CHUNK, n - off);
    HIPCHK(hipMemcpyAsync(S.buf[k % STAGE_RING], d_src + off, len, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipEventRecord(S.ev[k % STAGE_RING], stream));
    return ZES_OK;
  };
  for (uint32_t k = 0; k < chunks && k < (uint32_t)STAGE_RING - 1; k++)
    if ((rc = issue(k))) return rc;
  for (uint32_t k = 0; k < chunks; k++) {
    if (k + STAGE_RING - 1 < chunks && (rc = issue(k + STAGE_RING - 1))) return rc;  // its slot was emptied in the round before
    HIPCHK(hipEventSynchronize(S.ev[k % STAGE_RING]));
    const uint64_t off = (uint64_t)k * STAGE_CHUNK;
    S.pool.copy(dst + off, S.buf[k % STAGE_RING], (size_t)std::min<uint64_t>(STAGE_CHUNK, n - off));
  }
  return ZES_OK;
}

// ---- kernel timing (HIP events on the library's stream) ----
// events are pooled: creating and destroying a pair per launch costs more than recording them
hipEvent_t take_event() {
  if (!g.event_pool.empty()) {
    hipEvent_t e = g.event_pool.back();
    g.event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

struct Timed {
  hipEvent_t a = nullptr, b = nullptr;
  const char* name;
  hipStream_t st;
  explicit Timed(const char* n, hipStream_t stream = nullptr) : name(n), st(stream ? stream : g.stream) {
    if (g.profiling) {
      a = take_event();
      b = take_event();
      (void)hipEventRecord(a, st);
    }
  }
  ~Timed() {
    if (g.profiling) {
      (void)hipEventRecord(b, st);
      g.pending.push_back({name, {a, b}});
    }
    // ZES_TRACE_KERNELS: wait for the launch and name it on stderr (the kernel after the last name printed is the one
    // a GPU fault belongs to)
    static const bool trace = getenv("ZES_TRACE_KERNELS") != nullptr;
    if (trace) {
      const hipError_t e = hipStreamSynchronize(g.stream);
      fprintf(stderr, "zes kernel done: %s (%s)\n", name, hipGetErrorName(e));
      fflush(stderr);
    }
  }
};

void collect_times() {
  if (!g.profiling) return;
  std::map<std::string, KTime> acc;
  std::vector<std::string> order;
  for (auto& p : g.pending) {
    float ms = 0;
    (void)hipEventSynchronize(p.second.second);
    (void)hipEventElapsedTime(&ms, p.second.first, p.second.second);
    if (!acc.count(p.first)) order.push_back(p.first);
    acc[p.first].ms += ms;
    acc[p.first].launches++;
    g.event_pool.push_back(p.second.first);
    g.event_pool.push_back(p.second.second);
  }
  g.pending.clear();
  g.last_times.clear();
  for (auto& n : order) g.last_times.push_back({n, acc[n]});
}

// ---- deflate ----
bool deflate_throws(uint64_t n) { return n == 0 || n == 1 || (n % ZES_BLK) == 1; }
uint64_t deflate_bound(uint64_t n) { return ((n < ZES_BLK / 2) ? (uint64_t)ZES_BLK : n * 2) + 6; }

// Core: count buffers inside d_in / d_out.  Buffers whose status[] comes back non-zero were
// rejected on the host (throw cases, capacity) and are skipped by the device pass.
int deflate_batch_core(const uint8_t* d_in, const uint64_t* in_off, const uint64_t* in_len, uint8_t* d_out,
                       const uint64_t* out_off, const uint64_t* out_cap, uint64_t* out_len, int32_t* status,
                       uint32_t count, const uint64_t* in_read = nullptr, const uint32_t* bflags = nullptr, uint32_t* adler_out = nullptr,
                       const uint32_t* start_bits = nullptr, bool defer = false) {
  std::vector<ZesBuf> hb;
  uint64_t nblk_total = 0;
  std::vector<uint32_t> live;
  hb.reserve(count);
  for (uint32_t i = 0; i < count; i++) {
    out_len[i] = 0;
    if (deflate_throws(in_len[i])) {
      status[i] = ZES_E_CORRUPT;
      continue;
    }
    if (out_cap[i] < deflate_bound(in_len[i])) {
      status[i] = ZES_E_NOSPACE;
      out_len[i] = deflate_bound(in_len[i]);
      continue;
    }
    if ((out_off[i] & 15u) || (((uintptr_t)d_out) & 15u)) {
      status[i] = ZES_E_ARG;
      continue;
    }
    status[i] = ZES_OK;
    ZesBuf b;
    b.in_off = in_off[i];
    b.n = in_len[i];
    b.out_off = out_off[i];
    b.cap = out_cap[i];
    b.first_blk = (uint32_t)nblk_total;
    b.nblk = (uint32_t)((in_len[i] + ZES_BLK - 1) / ZES_BLK);
    b.n_read = in_read ? std::max(in_read[i], in_len[i]) : in_len[i];
    b.flags = bflags ? bflags[i] : 0u;
    b.start_bit = start_bits ? start_bits[i] : 0u;
    nblk_total += b.nblk;
    hb.push_back(b);
    live.push_back(i);
  }
  if (hb.empty()) return ZES_OK;
  if (nblk_total >= (1ull << 31)) return ZES_E_ARG;
  const uint32_t nbuf = (uint32_t)hb.size(), nblk = (uint32_t)nblk_total;
  int rc;
  if ((rc = ensure(g.bufs, sizeof(ZesBuf) * nbuf))) return rc;
  if ((rc = ensure(g.blks, sizeof(ZesBlk) * nblk))) return rc;
  if ((rc = ensure(g.idx_a, (size_t)nblk * ZES_BLK * 4))) return rc;
  if ((rc = ensure(g.idx_b, (size_t)nblk * ZES_BLK * 4))) return rc;
  if ((rc = ensure(g.sdelta, (size_t)nblk * ZES_BLK * 2 + 64))) return rc;
  if ((rc = ensure(g.tmask, (size_t)nblk * ZES_TMASK_WORDS * 4))) return rc;  // k_lz_match_lazy -> k_lz_parse: the chain's positions
  if ((rc = ensure(g.mlist, (size_t)nblk * ZES_MLIST_WORDS * 4))) return rc;  // k_lz_match -> k_lz_parse: the matches of a match-poor block
  if ((rc = ensure(g.hists, (size_t)nblk * 320 * 4))) return rc;
  if ((rc = ensure(g.codes, (size_t)nblk * 320 * 4))) return rc;
  if ((rc = ensure(g.hdrs, (size_t)nblk * ZES_HDR_WORDS * 4))) return rc;
  if ((rc = ensure(g.adler, (size_t)nbuf * 16))) return rc;
  if ((rc = ensure(g.res, sizeof(ZesRes) * nbuf))) return rc;
  // the buffer table goes up through pinned memory (a one-buffer call passes its entry as a kernel
  // argument instead); the block records and the cleared Adler accumulators are made on the device
  ZesBuf* dbufs = (ZesBuf*)g.bufs.p;
  ZesBlk* dblks = (ZesBlk*)g.blks.p;
  uint32_t* idx_a = (uint32_t*)g.idx_a.p;
  uint32_t* idx_b = (uint32_t*)g.idx_b.p;
  unsigned long long* adler = (unsigned long long*)g.adler.p;
  if (nbuf > 1) {
    if (sizeof(ZesBuf) * nbuf <= g.pinned_cap - PIN_UP) {
      memcpy((uint8_t*)g.pinned + PIN_UP, hb.data(), sizeof(ZesBuf) * nbuf);
      HIPCHK(hipMemcpyAsync(g.bufs.p, (uint8_t*)g.pinned + PIN_UP, sizeof(ZesBuf) * nbuf, hipMemcpyHostToDevice, g.stream));
    } else {
      HIPCHK(hipMemcpy(g.bufs.p, hb.data(), sizeof(ZesBuf) * nbuf, hipMemcpyHostToDevice));
    }
  }
  {
    Timed t("k_make_blks");
    const uint32_t nthr = std::max(nblk, 2u * nbuf);
    hipLaunchKernelGGL(k_make_blks, dim3((nthr + 255) / 256), dim3(256), 0, g.stream, hb[0], nbuf, dbufs, dblks, nblk, adler);
  }
  // Adler-32 (src/adler32.ts:1-10) needs the input and the cleared accumulators, and nobody but k_layout needs its
  // result: it runs on a stream of its own beside the LZ77 kernels — an HBM-bound pass of 64-byte-LDS workgroups next to
  // kernels that hold a block in LDS and wait on it (0.04 ms per 64 MiB off the critical path).
  {
    HIPCHK(hipEventRecord(g.ev_a0, g.stream));
    HIPCHK(hipStreamWaitEvent(g.s_adler, g.ev_a0, 0));
    Timed t("k_adler", g.s_adler);
    if (nbuf == 1) {  // one buffer: 64 KiB chunks, twice the workgroups
      const uint32_t nch = (uint32_t)((hb[0].n + ADLER_CHUNK - 1) / ADLER_CHUNK);
      hipLaunchKernelGGL(k_adler, dim3(nch), dim3(ADLER_THREADS), 0, g.s_adler, d_in, hb[0].in_off, hb[0].n, adler);
    } else {
      hipLaunchKernelGGL(k_adler_blocks, dim3(nblk), dim3(ADLER_THREADS), 0, g.s_adler, d_in, dbufs, dblks, adler);
    }
  }
  HIPCHK(hipEventRecord(g.ev_a1, g.s_adler));
  const bool sort_dbg = getenv("ZES_DEBUG_PHASES") != nullptr;
  if (sort_dbg) {
    int rc2 = ensure(g.dbg, (size_t)nblk * 64);
    if (rc2) return rc2;
    HIPCHK(hipMemsetAsync(g.dbg.p, 0, (size_t)nblk * 64, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    zes_sort_set_dbg((unsigned long long*)g.dbg.p);
  }
  // dense blocks (text, periodic data) get their index from k_lz_index (LDS-resident class sorts); a block it cannot take
  // goes back to k_lz_sort in a second launch that every other block leaves at once
  static const bool use_index = getenv("ZES_NO_INDEX") == nullptr;
  {
    Timed t("k_lz_sort");
    hipLaunchKernelGGL(k_lz_sort, dim3(nblk), dim3(SORT_THREADS), 0, g.stream, d_in, dbufs, dblks, idx_a, idx_b, (uint32_t*)g.idx_a.p, (uint16_t*)g.sdelta.p,
                       ZES_SORT_MODE_FIRST | (use_index ? ZES_SORT_USE_INDEX : 0u));
  }
  if (use_index) {
    {
      Timed t("k_lz_index");
      hipLaunchKernelGGL(k_lz_index, dim3(nblk), dim3(IDX_THREADS), 0, g.stream, d_in, dbufs, dblks, idx_a, idx_b, (uint32_t*)g.idx_a.p, (uint16_t*)g.sdelta.p);
    }
    Timed t("k_lz_sort_redo");
    hipLaunchKernelGGL(k_lz_sort, dim3(nblk), dim3(SORT_THREADS), 0, g.stream, d_in, dbufs, dblks, idx_a, idx_b, (uint32_t*)g.idx_a.p, (uint16_t*)g.sdelta.p,
                       ZES_SORT_MODE_REDO);
  }
  if (sort_dbg) {  // average shader-clock cycles per step of k_lz_sort
    HIPCHK(hipStreamSynchronize(g.stream));
    zes_sort_set_dbg(nullptr);
    std::vector<unsigned long long> h((size_t)nblk * 8);
    HIPCHK(hipMemcpy(h.data(), g.dbg.p, h.size() * 8, hipMemcpyDeviceToHost));
    double acc[8] = {0};
    uint32_t n = 0;
    for (uint32_t i = 0; i < nblk; i++) {
      if (!h[(size_t)i * 8 + 7]) continue;
      n++;
      for (int k = 1; k < 8; k++) acc[k] += (double)(h[(size_t)i * 8 + k] - h[(size_t)i * 8 + k - 1]);
    }
    if (n)
      fprintf(stderr, "zes sort steps (avg cycles over %u blocks): zero %.0f count %.0f flag %.0f compact %.0f stage %.0f three passes %.0f sd/inv %.0f\n", n,
              acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, acc[6] / n, acc[7] / n);
  }
  {
    Timed t("k_lz_match");  // match words go to idx_b (free after the sort)
    hipLaunchKernelGGL(k_lz_match, dim3(nblk), dim3(MATCH_THREADS), 0, g.stream, d_in, dbufs, dblks, idx_a, idx_b, (uint32_t*)g.mlist.p);
  }
  if (sort_dbg) {
    HIPCHK(hipMemsetAsync(g.dbg.p, 0, (size_t)nblk * 64, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    zes_lazy_set_dbg((unsigned long long*)g.dbg.p);
  }
  {
    // a batch of unlike buffers: the blocks heaviest first (k_lz_order), so that the launch does not end on a tail of text blocks
    const uint32_t* order = nullptr;
    if (nbuf > 1 && nblk > 256) {
      if ((rc = ensure(g.order, (size_t)nblk * 4))) return rc;
      Timed t("k_lz_order");
      hipLaunchKernelGGL(k_lz_order, dim3(1), dim3(1024), 0, g.stream, (const uint32_t*)idx_a, nblk, (uint32_t*)g.order.p);
      order = (const uint32_t*)g.order.p;
    }
    Timed t("k_lz_match_lazy");  // the blocks k_lz_sort flagged (most positions kept); the others return at once
    hipLaunchKernelGGL(k_lz_match_lazy, dim3(nblk), dim3(MATCH_THREADS), 0, g.stream, d_in, dbufs, dblks, idx_a,
                       (const uint32_t*)g.idx_a.p, (const uint16_t*)g.sdelta.p, idx_b, (uint32_t*)g.tmask.p, (uint32_t*)g.mlist.p, order);
  }
  if (sort_dbg) {  // average shader-clock cycles per phase of k_lz_match_lazy
    HIPCHK(hipStreamSynchronize(g.stream));
    zes_lazy_set_dbg(nullptr);
    std::vector<unsigned long long> h((size_t)nblk * 8);
    HIPCHK(hipMemcpy(h.data(), g.dbg.p, h.size() * 8, hipMemcpyDeviceToHost));
    double acc[8] = {0};
    uint32_t n = 0;
    for (uint32_t i = 0; i < nblk; i++) {
      if (!h[(size_t)i * 8 + 5]) continue;
      n++;
      for (int k = 1; k < 6; k++) acc[k] += (double)(h[(size_t)i * 8 + k] - h[(size_t)i * 8 + k - 1]);
      acc[6] += (double)h[(size_t)i * 8 + 6];
      acc[7] += (double)h[(size_t)i * 8 + 7];
    }
    if (n)
      fprintf(stderr, "zes lazy match steps (avg cycles over %u blocks): stage %.0f tail %.0f window chains %.0f entry chains %.0f true chain (only unmerged blocks) %.0f | first wave, window chains: %.0f loop turns\n", n,
              acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, acc[6] / n);
  }
  if (sort_dbg) {
    HIPCHK(hipMemsetAsync(g.dbg.p, 0, (size_t)nblk * 64, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    zes_parse_set_dbg((unsigned long long*)g.dbg.p);
  }
  {
    Timed t("k_lz_parse_small");  // tokens go to idx_a (free after the match pass)
    // two launches over all blocks: the blocks with a chain mask or a short match list run two to a compute unit
    // (k_lz_parse_small), the others need the exit maps' 128 KiB; each kernel leaves the other's blocks at once
    hipLaunchKernelGGL(k_lz_parse_small, dim3(nblk), dim3(PARSE_THREADS), 0, g.stream, d_in, dbufs, dblks, idx_b, idx_a, (uint32_t*)g.hists.p,
                       (const uint32_t*)g.tmask.p, (const uint32_t*)g.mlist.p);
  }
  {
    Timed t("k_lz_parse");
    hipLaunchKernelGGL(k_lz_parse, dim3(nblk), dim3(PARSE_THREADS), 0, g.stream, d_in, dbufs, dblks, idx_b, idx_a, (uint32_t*)g.hists.p,
                       (const uint32_t*)g.tmask.p, (const uint32_t*)g.mlist.p);
  }
  if (sort_dbg) {  // average shader-clock cycles per phase of k_lz_parse
    HIPCHK(hipStreamSynchronize(g.stream));
    zes_parse_set_dbg(nullptr);
    std::vector<unsigned long long> h((size_t)nblk * 8);
    HIPCHK(hipMemcpy(h.data(), g.dbg.p, h.size() * 8, hipMemcpyDeviceToHost));
    double acc[8] = {0};
    uint32_t n = 0;
    for (uint32_t i = 0; i < nblk; i++) {
      if (!h[(size_t)i * 8 + 7]) continue;
      n++;
      for (int k = 1; k < 8; k++) acc[k] += (double)(h[(size_t)i * 8 + k] - h[(size_t)i * 8 + k - 1]);
    }
    if (n)
      fprintf(stderr, "zes parse steps (avg cycles over %u blocks): A %.0f B %.0f C %.0f D1 %.0f D2 %.0f D3 %.0f out %.0f\n", n,
              acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, acc[6] / n, acc[7] / n);
  }
  if (sort_dbg) {
    HIPCHK(hipMemsetAsync(g.dbg.p, 0, (size_t)nblk * 64, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    zes_huff_set_dbg((unsigned long long*)g.dbg.p);
  }
  ZesRes* res_direct = nullptr;  // the read-back area as the device sees it
  if (sizeof(ZesRes) * nbuf <= PIN_UP) {
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, g.pinned, 0) == hipSuccess) res_direct = (ZesRes*)dp;
    else (void)hipGetLastError();
  }
  {
    Timed t("k_huff");
    hipLaunchKernelGGL(k_huff, dim3(nblk), dim3(HUFF_THREADS_HOST), 0, g.stream, dblks, (const uint32_t*)g.hists.p,
                       (uint32_t*)g.codes.p, (uint32_t*)g.hdrs.p);
  }
  if (sort_dbg) {  // average shader-clock cycles per step of k_huff
    HIPCHK(hipStreamSynchronize(g.stream));
    zes_huff_set_dbg(nullptr);
    std::vector<unsigned long long> h((size_t)nblk * 8);
    HIPCHK(hipMemcpy(h.data(), g.dbg.p, h.size() * 8, hipMemcpyDeviceToHost));
    double acc[8] = {0};
    uint32_t n = 0;
    for (uint32_t i = 0; i < nblk; i++) {
      if (!h[(size_t)i * 8 + 7]) continue;
      n++;
      for (int k = 1; k < 8; k++) acc[k] += (double)(h[(size_t)i * 8 + k] - h[(size_t)i * 8 + k - 1]);
    }
    if (n)
      fprintf(stderr, "zes huff steps (avg cycles over %u blocks): lit/len lengths %.0f distance lengths %.0f codes %.0f run-length coding %.0f its code %.0f header bits %.0f totals+out %.0f\n", n,
              acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, acc[6] / n, acc[7] / n);
  }
  {
    HIPCHK(hipStreamWaitEvent(g.stream, g.ev_a1, 0));  // the checksums
    Timed t("k_layout");
    // (the results go straight into the page-locked read-back area when they fit it: no copy command behind the kernels)
    hipLaunchKernelGGL(k_layout, dim3(nbuf), dim3(256), 0, g.stream, d_out, dbufs, dblks, adler, res_direct ? res_direct : (ZesRes*)g.res.p);
  }
  {
    Timed t("k_emit");
    hipLaunchKernelGGL(k_emit, dim3(nblk), dim3(EMIT_THREADS), 0, g.stream, d_out, dbufs, dblks, idx_a,
                       (const uint32_t*)g.codes.p, (const uint32_t*)g.hdrs.p);
  }
  HIPCHK(hipGetLastError());
  if (sizeof(ZesRes) * nbuf > g.pinned_cap) {
    HIPCHK(hipHostFree(g.pinned));
    g.pinned_cap = sizeof(ZesRes) * nbuf * 2;
    HIPCHK(hipHostMalloc(&g.pinned, g.pinned_cap, hipHostMallocDefault));
  }
  if (!res_direct) HIPCHK(hipMemcpyAsync(g.pinned, g.res.p, sizeof(ZesRes) * nbuf, hipMemcpyDeviceToHost, g.stream));
  if (defer) return ZES_OK;  // (one buffer: the caller reads g.pinned after its own synchronisation, deflate_piece_finish)
  HIPCHK(hipStreamSynchronize(g.stream));
  collect_times();
  const ZesRes* r = (const ZesRes*)g.pinned;
  for (uint32_t k = 0; k < nbuf; k++) {
    out_len[live[k]] = r[k].out_len;
    status[live[k]] = r[k].status;
    if (adler_out) adler_out[live[k]] = r[k].aux;
  }
  return ZES_OK;
}

// ---- inflate ----
int read_res(ZesRes* out) {
  HIPCHK(hipMemcpyAsync(g.pinned, g.res.p, sizeof(ZesRes), hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  *out = *(const ZesRes*)g.pinned;
  return ZES_OK;
}

constexpr uint32_t INF_GROUP = 4096;  // buffers per T1 group (table and read-backs fit the pinned area)

struct InfJob {
  uint64_t in_off, c, out_off, cap;
  uint64_t out_len;
  int status;  // reference-equivalent status once done
  int tier;    // 0 = not decoded yet
  // a piece of a longer stream on its way through the segment-parallel tier (inflate_segments_pieces): where work item
  // 0 starts, how much output exists in front of out_off, "a chain that ends in front of the piece's last, cut block
  // is fine"; out: the bit behind the last block decoded (relative to in_off), whether it was the stream's final one
  uint32_t start0 = 16, hist = 0;
  bool partial = false, final_seen = false;
  uint64_t end_bit = 0;
  int btype0 = -1;  // BTYPE of the block at bit 16, when the block-parallel tier's scan has sent it along (-1: not known)
};

bool t1_eligible(const InfJob& j, uint32_t flags) { return !(flags & (ZES_F_NO_FASTPATH | ZES_F_PIECES)) && j.c >= 64 && j.c < (1ull << 29); }

// The header test of the block-start search, two launches: k_inf_verify (a lane per survivor, the first VERIFY_STEPS
// code-length symbols: nearly all survivors end there) and k_inf_verify_long (a wave per survivor still alive: the
// real headers, ~300 symbols each).  `total_c`: compressed bytes behind the survivors (sizes the grids).
int launch_verify(const uint8_t* d_in, const ZesInfBuf* dbufs, uint32_t surv_cap, uint32_t* counters, uint32_t* cnt, uint32_t loose,
                  uint64_t total_c, uint32_t div) {
  int rc;
  (void)div;
  // the list of the long pass: one survivor in ~350 bytes of stream, one in eight of them listed; ten times that
  const uint32_t vlong_cap = (uint32_t)std::min<uint64_t>(total_c / 256 + 4096, surv_cap);
  if ((rc = ensure(g.vlong, (size_t)vlong_cap * 24))) return rc;
  // one lane per survivor: waves for all of them at once (the loop in the kernel takes what is beyond)
  uint32_t nwg = (uint32_t)std::min<uint64_t>(total_c / 16384 + 1, 8192);
  if (const char* e = getenv("ZES_VERIFY_DIV")) nwg = (uint32_t)std::min<uint64_t>(total_c / (uint64_t)atoi(e) + 1, 8192);
  {
    Timed t("k_inf_verify");
    hipLaunchKernelGGL(k_inf_verify, dim3(nwg), dim3(64), 0, g.stream, d_in, dbufs, (const unsigned long long*)g.surv.p, surv_cap, counters,
                       (uint32_t*)g.cand.p, cnt, loose, (uint32_t*)g.vlong.p, vlong_cap,
                       getenv("ZES_VERIFY_STEPS") ? (uint32_t)atoi(getenv("ZES_VERIFY_STEPS")) : 32u);
  }
  {
    Timed t("k_inf_verify_long");
    // one wave per listed survivor (about one in 2800 bytes of stream)
    // (a wave per item, no second item for most waves: 64 MiB of random bytes list 26 000 — 0.076 -> 0.065 ms against a cap of 8192)
    const uint32_t nlong = (uint32_t)std::min<uint64_t>(total_c / 2048 + 1, 32768);
    hipLaunchKernelGGL(k_inf_verify_long, dim3(nlong), dim3(64), 0, g.stream, d_in, dbufs, (const unsigned long long*)g.surv.p, surv_cap,
                       counters, (uint32_t*)g.cand.p, cnt, loose, (const uint32_t*)g.vlong.p, vlong_cap);
  }
  if (getenv("ZES_VERIFY_DBG")) {
    uint32_t hc[4];
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipMemcpy(hc, counters, 16, hipMemcpyDeviceToHost));
    fprintf(stderr, "verify: survivors %u, handed to the wave form %u\n", hc[0], hc[1]);
  }
  return ZES_OK;
}

// ZES_DEBUG_HOSTLAPS: host-side time between the synchronisation points of an inflate call (which round trips a call pays)
void host_lap(const char* what) {
  static const bool on = getenv("ZES_DEBUG_HOSTLAPS") != nullptr;
  if (!on) return;
  static thread_local std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
  const auto now = std::chrono::steady_clock::now();
  fprintf(stderr, "zes host lap: %-34s %8.1f us\n", what, std::chrono::duration<double, std::micro>(now - last).count());
  last = now;
}

// ZES_DEBUG_PHASES: average shader-clock cycles per phase of the block decoder (k_inf_block_par*, k_inf_seg_block_par)
int print_par_phases(const unsigned long long* dbg, uint64_t work) {
    std::vector<unsigned long long> h((size_t)work * ZES_PAR_DBG_ROW);
    HIPCHK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    double acc[8] = {0}, t0[3] = {0}, t15[3] = {0}, fb_lanes = 0, fb_waves = 0, hs[5] = {0}, p4[5] = {0}, why[3] = {0};
    uint32_t cntd = 0;
    for (uint32_t i = 0; i < work; i++) {
      const unsigned long long* r = &h[(size_t)i * ZES_PAR_DBG_ROW];
      if (!r[7]) continue;
      cntd++;
      for (int k = 1; k < 8; k++) acc[k] += (double)(r[k] - r[k - 1]);
      for (int k = 0; k < 3; k++) {  // table-phase steps relative to the end of the header phase
        t0[k] += (double)(r[8 + k] - r[1]);
        t15[k] += (double)(r[12 + k] - r[1]);
      }
      hs[0] += (double)(r[16] - r[0]);
      for (int k = 1; k < 5; k++) hs[k] += (double)(r[16 + k] - r[15 + k]);
      for (int k = 0; k < 5; k++) p4[k] += (double)r[24 + k];
      fb_lanes += (double)r[11];
      fb_waves += (double)r[15];
      for (int k = 0; k < 3; k++) why[k] += (double)r[29 + k];
    }
    fprintf(stderr, "zes phases (avg cycles over %u blocks): hdr %.0f tables %.0f compose %.0f count %.0f emit %.0f resolve %.0f flush %.0f\n",
            cntd, acc[1] / cntd, acc[2] / cntd, acc[3] / cntd, acc[4] / cntd, acc[5] / cntd, acc[6] / cntd, acc[7] / cntd);
    fprintf(stderr, "zes table steps, cycles since the header: first wave window %.0f landing %.0f fill %.0f | last wave %.0f %.0f %.0f\n",
            t0[0] / cntd, t0[1] / cntd, t0[2] / cntd, t15[0] / cntd, t15[1] / cntd, t15[2] / cntd);
    fprintf(stderr, "zes header steps (avg cycles): staging %.0f fixed fields + code-length code %.0f code lengths %.0f lit/len tables %.0f distance tables %.0f\n",
            hs[0] / cntd, hs[1] / cntd, hs[2] / cntd, hs[3] / cntd, hs[4] / cntd);
    fprintf(stderr, "zes resolve steps (avg cycles): carry+clear %.0f fill %.0f jumping %.0f copy %.0f | %.1f barrier rounds per block\n", p4[0] / cntd,
            p4[1] / cntd, p4[2] / cntd, p4[3] / cntd, p4[4] / cntd);
    {
      double c0 = 0, c15 = 0;
      for (uint32_t i = 0; i < work; i++) {
        const unsigned long long* r = &h[(size_t)i * ZES_PAR_DBG_ROW];
        if (!r[7]) continue;
        c0 += (double)(r[22] - r[3]);
        c15 += (double)(r[23] - r[3]);
      }
      fprintf(stderr, "zes count pass, cycles since its start: first wave through %.0f, last wave %.0f\n", c0 / cntd, c15 / cntd);
    }
    fprintf(stderr, "zes 8-bit table path: %.2f lanes in %.2f waves per block fell back to the generic construction (segment shape or list full %.2f, three positions under one token %.2f, look-back %.2f)\n", fb_lanes / cntd, fb_waves / cntd, why[0] / cntd, why[1] / cntd, why[2] / cntd);
    return ZES_OK;
}

// T1 over a group of buffers: every launch covers all of them (scan, verify, sort, one decode work
// item per candidate block, chain check), two host synchronisations for the whole group.  Jobs the
// tier settles get tier = 1; the others are left for the per-buffer tiers.
int inflate_t1_group(const uint8_t* d_in, uint8_t* d_out, InfJob* jobs, const uint32_t* ids, uint32_t nbuf, bool check_first, uint32_t flags) {
  int rc;
  g.sv_ok = false;  // (g.surv is about to be rewritten)
  ZesInfBuf* hb = (ZesInfBuf*)((uint8_t*)g.pinned + PIN_UP);
  uint64_t chunks = 0, cands = 0, total_c = 0;
  for (uint32_t i = 0; i < nbuf; i++) {
    const InfJob& j = jobs[ids[i]];
    ZesInfBuf& b = hb[i];
    b.in_off = j.in_off;
    b.c = j.c;
    b.out_off = j.out_off;
    b.cap = j.cap;
    b.first_chunk = (uint32_t)chunks;
    b.cand_base = (uint32_t)cands;
    // a reference-made stream has one block per 131072 bytes of output: more candidates than the caller's capacity
    // has blocks (plus room for false ones) means another encoder wrote the stream — counted as an overflow, and the
    // sort never sees more than this many (a zlib stream with 250-byte blocks has 250 000 of them)
    b.cand_cap = (uint32_t)std::min<uint64_t>(j.c / 64 + 64, j.cap / ZES_BLK + 65);
    b.work_first = 0;
    b.start_rel = 0;
    b.own_rel = 0xFFFFFFFFu;
    b.range_flags = 0;
    b.pad = 0;
    chunks += (j.c + INF_SCAN_BYTES - 1) / INF_SCAN_BYTES;
    cands += b.cand_cap;
    total_c += j.c;
  }
  if (chunks >= (1ull << 31) || cands >= (1ull << 31)) return ZES_OK;  // leave the jobs to the per-buffer tiers
  memset(&hb[nbuf], 0, sizeof(ZesInfBuf));
  hb[nbuf].first_chunk = (uint32_t)chunks;
  hb[nbuf].cand_base = (uint32_t)cands;
  if (nbuf == 1) {  // see the single-synchronisation path below
    hb[1].work_first = ZES_WORK_AUTO;
    hb[1].cand_cap = hb[0].cand_cap;
  }
  const uint32_t surv_cap = (uint32_t)std::min<uint64_t>(total_c / 4 + 1024ull * nbuf, 1ull << 30);
  const size_t cnt_bytes = 16 + (size_t)nbuf * 4 + (((size_t)nbuf + 3) & ~(size_t)3);  // counters[4], cnt[nbuf], first bytes [nbuf]
  if ((rc = ensure(g.ibufs, sizeof(ZesInfBuf) * (nbuf + 1)))) return rc;
  if ((rc = ensure(g.surv, (size_t)surv_cap * 8))) return rc;
  if ((rc = ensure(g.cand, (size_t)cands * 4))) return rc;
  if ((rc = ensure(g.cand_sorted, (size_t)cands * 4))) return rc;
  if ((rc = ensure(g.cres, sizeof(ZesCandRes) * cands))) return rc;
  if ((rc = ensure(g.map, (size_t)cands * 4))) return rc;
  if ((rc = ensure(g.counters, cnt_bytes))) return rc;
  if ((rc = ensure(g.res, sizeof(ZesRes) * nbuf))) return rc;
  const ZesInfBuf* dbufs = (const ZesInfBuf*)g.ibufs.p;
  uint32_t* counters = (uint32_t*)g.counters.p;
  uint32_t* cnt = counters + 4;
  uint8_t* dfirst = (uint8_t*)(cnt + nbuf);
  if (nbuf == 1) {  // table and cleared counters straight from kernel arguments
    hipLaunchKernelGGL(k_inf_set_table1, dim3(1), dim3(64), 0, g.stream, hb[0], hb[1], (ZesInfBuf*)g.ibufs.p, counters,
                       (uint32_t)(cnt_bytes / 4));
  } else {
    HIPCHK(hipMemcpyAsync(g.ibufs.p, hb, sizeof(ZesInfBuf) * (nbuf + 1), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemsetAsync(g.counters.p, 0, cnt_bytes, g.stream));
  }
  {
    Timed t("k_inf_scan");
    hipLaunchKernelGGL(k_inf_scan, dim3((uint32_t)chunks), dim3(INF_SCAN_THREADS), 0, g.stream, d_in, dbufs, nbuf,
                       (unsigned long long*)g.surv.p, surv_cap, counters, dfirst, (flags & ZES_F_LOOSE_CANDIDATES) ? 1u : 2u,
                       (const uint8_t*)g.kraft.p);
  }
  // (k_inf_verify, measured on 64 MiB: 8192 workgroups 0.33 ms, 2048 0.26 ms, 512 0.36 ms — about one survivor in 256 input
  // bytes, and a lane should get a few of them)
  if ((rc = launch_verify(d_in, dbufs, surv_cap, counters, cnt, (flags & ZES_F_LOOSE_CANDIDATES) ? 1u : 0u, total_c, 32768))) return rc;
  // One buffer: nothing has to come back before the decode is launched.  The grid is sized for the most
  // blocks the caller's capacity can hold (plus room for false candidates); the kernels take the real
  // candidate count from device memory (table sentinel ZES_WORK_AUTO) and the host reads counters and
  // result together — one synchronisation per call.  Several buffers: the counts come back first.
  const bool one = nbuf == 1;
  uint32_t* hc = (uint32_t*)g.pinned;
  ZesRes* hres = (ZesRes*)((uint8_t*)g.pinned + 128 * 1024);
  uint64_t work = 0;
  std::vector<uint32_t> ncand(nbuf);
  if (!one) {
    HIPCHK(hipMemcpyAsync(hc, g.counters.p, cnt_bytes, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));  // the table upload has completed too: hb may be rewritten
    for (uint32_t i = 0; i < nbuf; i++) {
      const uint8_t fb = ((const uint8_t*)(hc + 4 + nbuf))[i];
      if (fb & 0x40u) jobs[ids[i]].btype0 = (fb >> 4) & 3;
    }
    if (check_first) {  // CM nibble of the first byte (src/zlib.ts:13-16): the scan kernel sent it along
      const uint8_t* hfirst = (const uint8_t*)(hc + 4 + nbuf);
      for (uint32_t i = 0; i < nbuf; i++)
        if ((hfirst[i] & 15u) != 8u) {
          jobs[ids[i]].status = ZES_E_NOT_DEFLATE;
          jobs[ids[i]].tier = -1;
          hc[4 + i] = 0;  // no candidates are looked at
        }
    }
    const uint32_t nsurv0 = hc[0];
    if (nsurv0 == 0 || nsurv0 > surv_cap) return ZES_OK;  // nothing that looks like this format (or a poisoned count)
    for (uint32_t i = 0; i < nbuf; i++) {
      ncand[i] = hc[4 + i];
      hb[i].work_first = (uint32_t)work;
      if (ncand[i] > 0 && ncand[i] <= hb[i].cand_cap) work += ncand[i];
    }
    hb[nbuf].work_first = (uint32_t)work;
    if (work == 0) return ZES_OK;
    HIPCHK(hipMemcpyAsync(g.ibufs.p, hb, sizeof(ZesInfBuf) * (nbuf + 1), hipMemcpyHostToDevice, g.stream));
  } else {
    work = hb[1].cand_cap;  // the launch bound written into the sentinel above
  }
  bool direct = false, hostchain = false;
  ZesParMirror mir{};
  unsigned long long* dbg = nullptr;
  if (getenv("ZES_DEBUG_PHASES")) {
    if ((rc = ensure(g.dbg, (size_t)work * ZES_PAR_DBG_ROW * 8))) return rc;
    HIPCHK(hipMemsetAsync(g.dbg.p, 0, (size_t)work * ZES_PAR_DBG_ROW * 8, g.stream));
    dbg = (unsigned long long*)g.dbg.p;
  }
  // (the candidates are ranked by the block decoder itself: every work item finds its own rank and the two candidates
  // behind it in the unsorted list, and leaves the sorted list for the chain check)
  {
    // compressible data (the streams are shorter than 0.7 of the room for their outputs): the variant whose transfer
    // tables look at two windows; incompressible data runs ~4 % faster in the smaller kernel
    uint64_t total_cap = 0;
    for (uint32_t i = 0; i < nbuf; i++) total_cap += jobs[ids[i]].cap;
    const bool two = total_c * 10 < total_cap * 7;
    Timed t(two ? "k_inf_block_par2" : "k_inf_block_par");
    auto kern = two ? k_inf_block_par2 : k_inf_block_par;
    // One buffer and a launch bound the mirror area holds: every work item also puts its result and its block's start bit
    // into page-locked host memory, work item 0 the counters, and the HOST follows the chain after the one synchronisation —
    // what k_inf_chain does, on a few hundred 16-byte records: no chain kernel behind this one (11 us + a kernel boundary of
    // a 0.8 ms call).  Only a chain that needs the slots moved (false candidates between the blocks) still runs that kernel,
    // for its map.
    if (one && work <= MIRROR_ITEMS && g.mirror) {
      void* dm = nullptr;
      void* dp0 = nullptr;
      if (hipHostGetDevicePointer(&dm, g.mirror, 0) == hipSuccess && hipHostGetDevicePointer(&dp0, g.pinned, 0) == hipSuccess) {
        mir.cres_host = (ZesCandRes*)dm;
        mir.start_host = (uint32_t*)((uint8_t*)dm + MIRROR_ITEMS * sizeof(ZesCandRes));
        mir.counters = counters;
        mir.counters_host = (uint32_t*)dp0;
        mir.counter_words = (uint32_t)(cnt_bytes / 4);
        hostchain = true;
      } else {
        (void)hipGetLastError();
      }
    }
    hipLaunchKernelGGL(kern, dim3((uint32_t)work), dim3(PAR_THREADS), 0, g.stream, d_in, d_out, dbufs, nbuf,
                       (const uint32_t*)cnt, (const uint32_t*)g.cand_sorted.p, (const uint32_t*)nullptr, (ZesCandRes*)g.cres.p, dbg, (const uint32_t*)nullptr,
                       (const uint32_t*)g.cand.p, (uint32_t*)g.cand_sorted.p, mir);
  }
  auto device_chain = [&]() -> int {
    Timed t("k_inf_chain");
    // one buffer: the kernel puts its result and the counters straight into the page-locked read-back area (no copy
    // commands behind the kernels: ~10 us of a 0.8 ms call)
    void* dp = nullptr;
    if (one && hipHostGetDevicePointer(&dp, g.pinned, 0) != hipSuccess) {
      (void)hipGetLastError();
      dp = nullptr;
    }
    direct = one && dp != nullptr;
    hipLaunchKernelGGL(k_inf_chain, dim3(nbuf), dim3(256), 0, g.stream, dbufs, (const uint32_t*)cnt, (const uint32_t*)g.cand_sorted.p,
                       (const ZesCandRes*)g.cres.p, (const uint32_t*)nullptr, (uint32_t*)g.map.p,
                       direct ? (ZesRes*)((uint8_t*)dp + 128 * 1024) : (ZesRes*)g.res.p, (const uint32_t*)counters, (uint32_t)(cnt_bytes / 4),
                       direct ? (uint32_t*)dp : (uint32_t*)nullptr);
    if (!direct) {
      if (one) HIPCHK(hipMemcpyAsync(hc, g.counters.p, cnt_bytes, hipMemcpyDeviceToHost, g.stream));
      HIPCHK(hipMemcpyAsync(hres, g.res.p, sizeof(ZesRes) * nbuf, hipMemcpyDeviceToHost, g.stream));
    }
    return ZES_OK;
  };
  if (!hostchain && (rc = device_chain())) return rc;
  host_lap("(work before the block-parallel tier)");
  HIPCHK(hipStreamSynchronize(g.stream));
  host_lap("T1: search + decode + chain");
  if (hostchain) {
    // k_inf_chain's walk on the mirror (one buffer; start[k] = the bit candidate k's block starts at, rank order)
    const ZesCandRes* hcr = (const ZesCandRes*)g.mirror;
    const uint32_t* hst = (const uint32_t*)((const uint8_t*)g.mirror + MIRROR_ITEMS * sizeof(ZesCandRes));
    ZesRes r;
    r.status = 1;
    r.out_len = 0;
    r.aux = 0;
    const uint32_t cnt0 = hc[4];
    const uint32_t nc = std::min(cnt0, hb[0].cand_cap);
    const uint32_t nwork = std::min<uint32_t>(nc, (uint32_t)work);
    if (hc[0] != 0 && hc[0] <= surv_cap && nc != 0 && cnt0 <= hb[0].cand_cap && nc <= nwork && hst[0] == 16u + hb[0].start_rel) {
      uint32_t K = 0xFFFFFFFFu;
      for (uint32_t k = 0; k < nwork; k++)
        if ((hcr[k].flags & 3u) == 3u) {
          K = k;
          break;
        }
      bool fast = K != 0xFFFFFFFFu;
      uint64_t total = 0;
      for (uint32_t k = 0; fast && k <= K; k++) {
        const ZesCandRes& c = hcr[k];
        if (!(c.flags & 1u)) fast = false;
        total += c.out_len;
        if (k < K && (c.out_len != ZES_BLK || k + 1 >= nc || (uint64_t)hst[k + 1] != c.end_bit)) fast = false;
      }
      if (fast) {
        r.status = 0;
        r.out_len = total;
        r.aux = 1;
      } else {  // false candidates between the blocks? follow end bit -> next start
        uint32_t j = 0, k = 0;
        total = 0;
        bool ok = false;
        for (;;) {
          const ZesCandRes& c = hcr[j];
          if (!(c.flags & 1u)) break;
          k++;
          total += c.out_len;
          if (c.flags & 2u) {
            ok = true;
            break;
          }
          if (c.out_len != ZES_BLK) break;
          const uint32_t* lo = std::lower_bound(hst + j + 1, hst + nc, (uint64_t)c.end_bit, [](uint32_t a, uint64_t b) { return (uint64_t)a < b; });
          if (lo == hst + nc || (uint64_t)*lo != c.end_bit) break;
          j = (uint32_t)(lo - hst);
        }
        if (ok) {  // the slots are shifted: the chain kernel's map is needed (rare)
          if ((rc = device_chain())) return rc;
          HIPCHK(hipStreamSynchronize(g.stream));
          r = hres[0];
          (void)k;
        }
      }
    }
    if (r.status != 2) hres[0] = r;
    if (getenv("ZES_DEBUG")) {
      fprintf(stderr, "zes T1 host chain: nsurv %u cnt %u cap %u work %llu -> status %d out_len %llu | start0 %u", hc[0], hc[4], hb[0].cand_cap,
              (unsigned long long)work, r.status, (unsigned long long)r.out_len, hst[0]);
      for (uint32_t k = 0; k < std::min<uint32_t>(hc[4], 10u); k++)
        fprintf(stderr, " [%u: start %u end %llu len %u fl %u]", k, hst[k], (unsigned long long)hcr[k].end_bit, hcr[k].out_len, hcr[k].flags);
      fprintf(stderr, "\n");
    }
  }
  const uint32_t nsurv = hc[0];
  if (one) {
    if (((const uint8_t*)(hc + 5))[0] & 0x40u) jobs[ids[0]].btype0 = (((const uint8_t*)(hc + 5))[0] >> 4) & 3;
    if (check_first && (((const uint8_t*)(hc + 5))[0] & 15u) != 8u) {  // src/zlib.ts:13-16
      jobs[ids[0]].status = ZES_E_NOT_DEFLATE;
      jobs[ids[0]].tier = -1;
      return ZES_OK;
    }
    ncand[0] = hc[4];
    if (nsurv != 0 && nsurv <= surv_cap && !(flags & ZES_F_LOOSE_CANDIDATES)) {  // the scan ran to its end and its list is whole
      g.sv_ok = true;
      g.sv_din = d_in;
      g.sv_list = g.surv.p;
      g.sv_in_off = jobs[ids[0]].in_off;
      g.sv_c = jobs[ids[0]].c;
      g.sv_n = nsurv;
    }
    // nothing that looks like this format, a poisoned count, or more candidates than were launched
    if (nsurv == 0 || nsurv > surv_cap || ncand[0] == 0 || ncand[0] > hb[0].cand_cap || ncand[0] > work) return ZES_OK;
    hb[0].work_first = 0;
    work = ncand[0];
  }
  std::vector<ZesRes> r1(hres, hres + nbuf);
  if (dbg && (rc = print_par_phases(dbg, work))) return rc;
  if (getenv("ZES_DEBUG")) {
    for (uint32_t i = 0, shown_b = 0; i < nbuf && shown_b < 4; i++) {
      if (r1[i].status == 0) continue;
      shown_b++;
      const uint32_t nc = std::min(ncand[i], hb[i].cand_cap);
      fprintf(stderr, "zes T1: buf %u c=%llu nsurv(all)=%u ncand=%u chain status=%d aux=%u out_len=%llu\n", ids[i],
              (unsigned long long)hb[i].c, nsurv, ncand[i], r1[i].status, r1[i].aux, (unsigned long long)r1[i].out_len);
      if (r1[i].status != 1 || nc == 0) continue;
      std::vector<ZesCandRes> hcr(nc);
      std::vector<uint32_t> hcand(nc);
      HIPCHK(hipMemcpy(hcr.data(), (const ZesCandRes*)g.cres.p + hb[i].cand_base, sizeof(ZesCandRes) * nc, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(hcand.data(), (const uint32_t*)g.cand_sorted.p + hb[i].cand_base, 4 * (size_t)nc, hipMemcpyDeviceToHost));
      int shown = 0;
      for (uint32_t k = 0; k < nc && shown < 6; k++) {
        const bool chain_ok = (k + 1 == nc) || ((uint64_t)hcand[k + 1] + 16 == hcr[k].end_bit);
        if (!(hcr[k].flags & 1u) || !chain_ok || (hcr[k].out_len != ZES_BLK && k + 1 != nc)) {
          fprintf(stderr, "  cand %u start=%u end_bit=%llu next_start=%u out_len=%u flags=%u\n", k, hcand[k] + 16,
                  (unsigned long long)hcr[k].end_bit, k + 1 < nc ? hcand[k + 1] + 16 : 0, hcr[k].out_len, hcr[k].flags);
          shown++;
        }
      }
    }
  }
  // Buffers whose candidate list holds false positives between the blocks: every true block decoded fine, but
  // the blocks behind a false candidate sit one (or more) slots too far right.  The chain kernel left the true
  // chain in map[] (true block k = candidate map[k]) and has checked it block by block, so the blocks only
  // have to move: through a scratch copy, because sources and destinations overlap.  A block whose slot was cut
  // off by the caller's capacity (typically the last one) is decoded again, straight into its own slot.
  for (uint32_t i = 0; i < nbuf; i++) {
    if (r1[i].status != 2) continue;
    const uint32_t K = r1[i].aux;
    InfJob& j = jobs[ids[i]];
    std::vector<uint32_t> hmap(K);
    HIPCHK(hipMemcpyAsync(hmap.data(), (const uint32_t*)g.map.p + hb[i].cand_base, (size_t)K * 4, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    std::vector<uint32_t> mv_src, mv_dst, redo;
    for (uint32_t k = 0; k < K; k++) {
      if (hmap[k] == k) continue;
      if ((uint64_t)(hmap[k] + 1u) * ZES_BLK <= j.cap) {
        mv_src.push_back(hmap[k]);
        mv_dst.push_back(k);
      } else {
        redo.push_back(k);
      }
    }
    const uint32_t nmv = (uint32_t)mv_src.size(), nre = (uint32_t)redo.size();
    if ((rc = ensure(g.mvlist, (size_t)(3 * nmv + nre + 4) * 4))) return rc;
    uint32_t* dl = (uint32_t*)g.mvlist.p;  // [src slots][dst slots][0..nmv)[redo]
    if (nmv) {
      if ((rc = ensure(g.scratch, (size_t)nmv * ZES_BLK))) return rc;
      std::vector<uint32_t> up(3 * (size_t)nmv);
      for (uint32_t q = 0; q < nmv; q++) {
        up[q] = mv_src[q];
        up[nmv + q] = mv_dst[q];
        up[2 * (size_t)nmv + q] = q;
      }
      HIPCHK(hipMemcpy(dl, up.data(), up.size() * 4, hipMemcpyHostToDevice));
      Timed t("k_inf_move_slots");
      uint8_t* outb = d_out + j.out_off;
      hipLaunchKernelGGL(k_inf_move_slots, dim3(nmv * 32u), dim3(256), 0, g.stream, (uint8_t*)g.scratch.p, (const uint8_t*)outb,
                         (const uint32_t*)(dl + 2 * (size_t)nmv), (const uint32_t*)dl, nmv);
      hipLaunchKernelGGL(k_inf_move_slots, dim3(nmv * 32u), dim3(256), 0, g.stream, outb, (const uint8_t*)g.scratch.p,
                         (const uint32_t*)(dl + nmv), (const uint32_t*)(dl + 2 * (size_t)nmv), nmv);
    }
    bool ok = true;
    if (nre) {
      HIPCHK(hipMemcpy(dl + 3 * (size_t)nmv, redo.data(), (size_t)nre * 4, hipMemcpyHostToDevice));
      // a two-entry table for this buffer alone: work item -> slot through redo[], K work items in all
      ZesInfBuf* one = (ZesInfBuf*)((uint8_t*)g.pinned + PIN_UP) + nbuf + 2;
      one[0] = hb[i];
      one[0].work_first = 0;
      one[1] = hb[i];
      one[1].work_first = K;
      if ((rc = ensure(g.ibufs2, sizeof(ZesInfBuf) * 2))) return rc;
      HIPCHK(hipMemcpyAsync(g.ibufs2.p, one, sizeof(ZesInfBuf) * 2, hipMemcpyHostToDevice, g.stream));
      {
        Timed t("k_inf_block_par");
        // cnt / candidates / map / results are indexed from this buffer's region: the table's cand_base does that
        hipLaunchKernelGGL(k_inf_block_par, dim3(nre), dim3(PAR_THREADS), 0, g.stream, d_in, d_out, (const ZesInfBuf*)g.ibufs2.p, 1u,
                           (const uint32_t*)cnt + i, (const uint32_t*)g.cand_sorted.p, (const uint32_t*)g.map.p,
                           (ZesCandRes*)g.cres.p, (unsigned long long*)nullptr, (const uint32_t*)(dl + 3 * (size_t)nmv), (const uint32_t*)nullptr,
                           (uint32_t*)nullptr, ZesParMirror{});
      }
      std::vector<ZesCandRes> hcr(nre);
      for (uint32_t q = 0; q < nre; q++)
        HIPCHK(hipMemcpyAsync(&hcr[q], (const ZesCandRes*)g.cres.p + hb[i].cand_base + redo[q], sizeof(ZesCandRes), hipMemcpyDeviceToHost,
                              g.stream));
      HIPCHK(hipStreamSynchronize(g.stream));
      for (uint32_t q = 0; q < nre; q++) {
        const bool last = redo[q] + 1u == K;
        ok = ok && (hcr[q].flags & 1u) && (last ? hcr[q].out_len <= ZES_BLK : hcr[q].out_len == ZES_BLK);
      }
    }
    r1[i].status = ok ? 0 : 1;  // out_len already holds the chain's total
  }
  for (uint32_t i = 0; i < nbuf; i++) {
    const ZesRes& r = r1[i];
    if (r.status != 0) continue;
    InfJob& j = jobs[ids[i]];
    j.tier = 1;
    j.out_len = r.out_len;
    j.status = r.out_len > j.cap ? ZES_E_NOSPACE : ZES_OK;
  }
  return ZES_OK;
}

// T1 over one PIECE of a reference-made stream: the blocks that start inside bits [lo_bit, own_bit) of the piece at
// d_in + in_off (c readable bytes: the piece and enough behind it for its last block, <= 144 KiB, and for the header
// of the block after it).  exact: the first block starts at lo_bit (known from the piece before); else the chain
// starts at the first block start found at or behind lo_bit.  Block k of the piece goes to d_out + out_off + k * 131072.
// handled = false: not a clean chain of reference-made blocks (the caller decodes the stream some other way).
struct RangeRes {
  bool handled = false;
  uint64_t out_len = 0, first_bit = 0, end_bit = 0;
  uint32_t nblocks = 0;
  bool final_block = false;
};
int inflate_t1_range(const uint8_t* d_in, uint64_t in_off, uint64_t c, uint64_t lo_bit, uint64_t own_bit, bool exact, uint8_t* d_out,
                     uint64_t out_off, uint64_t cap, uint32_t flags, RangeRes* rr) {
  int rc;
  *rr = RangeRes();
  if (c >= (1ull << 29) || lo_bit < 16) return ZES_OK;
  // a range without a block start (a piece in the middle of one block): nothing to decode, and that is an answer
  auto nothing_here = [&]() {
    if (exact) return ZES_OK;
    rr->handled = true;
    rr->first_bit = rr->end_bit = lo_bit;
    return ZES_OK;
  };
  if (c * 8 < lo_bit + 64) return nothing_here();
  ZesInfBuf* hb = (ZesInfBuf*)((uint8_t*)g.pinned + PIN_UP);
  memset(hb, 0, 2 * sizeof(ZesInfBuf));
  const uint64_t chunks = (c + INF_SCAN_BYTES - 1) / INF_SCAN_BYTES;
  hb[0].in_off = in_off;
  hb[0].c = c;
  hb[0].out_off = out_off;
  hb[0].cap = cap;
  hb[0].cand_cap = (uint32_t)std::min<uint64_t>(c / 64 + 64, 1ull << 23);  // (not from cap: a short output still gets its size)
  hb[0].start_rel = (uint32_t)(lo_bit - 16);
  hb[0].own_rel = (uint32_t)std::min<uint64_t>(own_bit >= 16 ? own_bit - 16 : 0, 0xFFFFFFFEull);
  hb[0].range_flags = exact ? 0u : ZES_START_ANY;
  hb[1].first_chunk = (uint32_t)chunks;
  hb[1].cand_base = hb[0].cand_cap;
  const uint32_t cands = hb[0].cand_cap;
  const uint32_t surv_cap = (uint32_t)std::min<uint64_t>(c / 4 + 1024ull, 1ull << 30);
  const size_t cnt_bytes = 16 + 4 + 4;  // counters[4], cnt[1], first byte
  if ((rc = ensure(g.ibufs, sizeof(ZesInfBuf) * 2))) return rc;
  if ((rc = ensure(g.surv, (size_t)surv_cap * 8))) return rc;
  if ((rc = ensure(g.cand, (size_t)cands * 4))) return rc;
  if ((rc = ensure(g.cand_sorted, (size_t)cands * 4))) return rc;
  if ((rc = ensure(g.cres, sizeof(ZesCandRes) * cands))) return rc;
  if ((rc = ensure(g.counters, cnt_bytes))) return rc;
  if ((rc = ensure(g.res, sizeof(ZesRes) * 2))) return rc;
  const ZesInfBuf* dbufs = (const ZesInfBuf*)g.ibufs.p;
  uint32_t* counters = (uint32_t*)g.counters.p;
  uint32_t* cnt = counters + 4;
  uint8_t* dfirst = (uint8_t*)(cnt + 1);
  HIPCHK(hipMemcpyAsync(g.ibufs.p, hb, sizeof(ZesInfBuf) * 2, hipMemcpyHostToDevice, g.stream));
  HIPCHK(hipMemsetAsync(g.counters.p, 0, cnt_bytes, g.stream));
  {
    Timed t("k_inf_scan");
    // (the scan's rule that a BFINAL position far from the end is no block start uses the end of the piece: a piece in
    // the middle of a stream merely keeps a few more survivors near its own end)
    g.sv_ok = false;
    hipLaunchKernelGGL(k_inf_scan, dim3((uint32_t)chunks), dim3(INF_SCAN_THREADS), 0, g.stream, d_in, dbufs, 1u, (unsigned long long*)g.surv.p,
                       surv_cap, counters, dfirst, (flags & ZES_F_LOOSE_CANDIDATES) ? 1u : 2u, (const uint8_t*)g.kraft.p);
  }
  if ((rc = launch_verify(d_in, dbufs, surv_cap, counters, cnt, (flags & ZES_F_LOOSE_CANDIDATES) ? 1u : 0u, c, 32768))) return rc;
  uint32_t* hc = (uint32_t*)g.pinned;
  HIPCHK(hipMemcpyAsync(hc, g.counters.p, cnt_bytes, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  const uint32_t nsurv = hc[0], ncand = hc[4];
  if (nsurv > surv_cap || ncand > hb[0].cand_cap) return ZES_OK;
  if (nsurv == 0 || ncand == 0) return nothing_here();
  hb[1].work_first = ncand;
  HIPCHK(hipMemcpyAsync(g.ibufs.p, hb, sizeof(ZesInfBuf) * 2, hipMemcpyHostToDevice, g.stream));
  {
    Timed t("k_inf_block_par");
    hipLaunchKernelGGL((c * 10 < cap * 7) ? k_inf_block_par2 : k_inf_block_par, dim3(ncand), dim3(PAR_THREADS), 0, g.stream, d_in, d_out, dbufs, 1u, (const uint32_t*)cnt,
                       (const uint32_t*)g.cand_sorted.p, (const uint32_t*)nullptr, (ZesCandRes*)g.cres.p, (unsigned long long*)nullptr, (const uint32_t*)nullptr,
                       (const uint32_t*)g.cand.p, (uint32_t*)g.cand_sorted.p, ZesParMirror{});
  }
  {
    Timed t("k_inf_chain");
    hipLaunchKernelGGL(k_inf_chain_range, dim3(1), dim3(256), 0, g.stream, dbufs, (const uint32_t*)cnt, (const uint32_t*)g.cand_sorted.p,
                       (const ZesCandRes*)g.cres.p, (ZesRes*)g.res.p, (unsigned long long*)nullptr);
  }
  ZesRes* hres = (ZesRes*)((uint8_t*)g.pinned + 128 * 1024);
  HIPCHK(hipMemcpyAsync(hres, g.res.p, sizeof(ZesRes) * 2, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  if (getenv("ZES_RANGE_DBG")) {
    std::vector<uint32_t> hcand(ncand);
    std::vector<ZesCandRes> hcr(ncand);
    HIPCHK(hipMemcpy(hcand.data(), g.cand_sorted.p, ncand * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hcr.data(), g.cres.p, ncand * sizeof(ZesCandRes), hipMemcpyDeviceToHost));
    fprintf(stderr, "range: c=%llu lo=%llu own=%llu exact=%d nsurv=%u ncand=%u status=%u\n", (unsigned long long)c, (unsigned long long)lo_bit,
            (unsigned long long)own_bit, (int)exact, nsurv, ncand, hres[0].status);
    for (uint32_t k = 0; k < ncand && k < 12; k++)
      fprintf(stderr, "  cand[%u]=%u flags=%u out_len=%llu end_bit=%llu\n", k, hcand[k], hcr[k].flags, (unsigned long long)hcr[k].out_len,
              (unsigned long long)hcr[k].end_bit);
  }
  if (hres[0].status != 0) return ZES_OK;
  rr->handled = true;
  rr->out_len = hres[0].out_len;
  rr->nblocks = hres[0].aux & 0x7FFFFFFFu;
  rr->final_block = (hres[0].aux >> 31) != 0;
  rr->end_bit = hres[1].out_len;
  rr->first_bit = hres[1].aux;
  return ZES_OK;
}

// The same in two halves, for the host call that decodes a stream piece by piece while its neighbours are on the link
// (inflate_host_pipelined): range_begin enqueues everything piece k needs — no look at a result in between: the
// block decoder is launched over a bound and takes the candidate count from device memory (ZES_WORK_AUTO), the piece's
// place in the output is the block count of the pieces before it, kept on the device (k_inf_chain_range adds to it,
// k_inf_set_table_range reads it) — and range_finish waits for the read-backs of that piece only.  The host enqueues
// piece k + 1 before it waits for piece k: the device never waits for the host between pieces.
struct RangePend {
  bool skip = false, nothing = false, exact = false;
  uint64_t lo_bit = 0;
  uint32_t surv_cap = 0, cand_cap = 0, bound = 0;
};
static uint32_t* range_hc(int slot) { return (uint32_t*)((uint8_t*)g.pinned + 96 * 1024 + slot * 64); }
static ZesRes* range_hres(int slot) { return (ZesRes*)((uint8_t*)g.pinned + 96 * 1024 + 256 + slot * 64); }
static unsigned long long* range_acc() { return (unsigned long long*)((uint8_t*)g.counters.p + 64); }
// scratch for pieces of up to cmax bytes, before anything is in flight (growing a buffer frees the old one)
int range_reserve(uint64_t cmax) {
  int rc;
  const uint32_t cands = (uint32_t)std::min<uint64_t>(cmax / 64 + 64, 1ull << 23);
  const uint32_t surv_cap = (uint32_t)std::min<uint64_t>(cmax / 4 + 1024ull, 1ull << 30);
  if ((rc = ensure(g.ibufs, sizeof(ZesInfBuf) * 2))) return rc;
  if ((rc = ensure(g.surv, (size_t)surv_cap * 8))) return rc;
  if ((rc = ensure(g.cand, (size_t)cands * 4))) return rc;
  if ((rc = ensure(g.cand_sorted, (size_t)cands * 4))) return rc;
  if ((rc = ensure(g.cres, sizeof(ZesCandRes) * cands))) return rc;
  if ((rc = ensure(g.counters, 128))) return rc;
  if ((rc = ensure(g.res, sizeof(ZesRes) * 2))) return rc;
  if ((rc = ensure(g.vlong, (size_t)std::min<uint64_t>(cmax / 256 + 4096, surv_cap) * 24))) return rc;
  HIPCHK(hipMemsetAsync(range_acc(), 0, 8, g.stream));
  return ZES_OK;
}
int range_begin(int slot, RangePend& pd, const uint8_t* d_in, uint64_t in_off, uint64_t c, uint64_t lo_bit, uint64_t own_bit, bool exact,
                uint8_t* d_out, uint64_t dcap, bool two, uint32_t flags) {
  int rc;
  pd = RangePend();
  pd.exact = exact;
  pd.lo_bit = lo_bit;
  if (c >= (1ull << 29) || lo_bit < 16) pd.skip = true;
  else if (c * 8 < lo_bit + 64) pd.nothing = true;
  if (!pd.skip && !pd.nothing) {
    ZesInfBuf b0, b1;
    memset(&b0, 0, sizeof b0);
    memset(&b1, 0, sizeof b1);
    const uint64_t chunks = (c + INF_SCAN_BYTES - 1) / INF_SCAN_BYTES;
    b0.in_off = in_off;
    b0.c = c;
    b0.cand_cap = (uint32_t)std::min<uint64_t>(c / 64 + 64, 1ull << 23);
    b0.start_rel = (uint32_t)(lo_bit - 16);
    b0.own_rel = (uint32_t)std::min<uint64_t>(own_bit >= 16 ? own_bit - 16 : 0, 0xFFFFFFFEull);
    b0.range_flags = exact ? 0u : ZES_START_ANY;
    pd.cand_cap = b0.cand_cap;
    pd.surv_cap = (uint32_t)std::min<uint64_t>(c / 4 + 1024ull, 1ull << 30);
    // the most blocks the output can hold, and room for false candidates
    pd.bound = (uint32_t)std::min<uint64_t>(b0.cand_cap, dcap / ZES_BLK + 65);
    b1.first_chunk = (uint32_t)chunks;
    b1.cand_base = b0.cand_cap;
    b1.cand_cap = pd.bound;
    b1.work_first = ZES_WORK_AUTO;
    const ZesInfBuf* dbufs = (const ZesInfBuf*)g.ibufs.p;
    uint32_t* counters = (uint32_t*)g.counters.p;
    uint32_t* cnt = counters + 4;
    uint8_t* dfirst = (uint8_t*)(cnt + 1);
    hipLaunchKernelGGL(k_inf_set_table_range, dim3(1), dim3(64), 0, g.stream, b0, b1, (ZesInfBuf*)g.ibufs.p, counters, 6u,
                       (const unsigned long long*)range_acc(), (unsigned long long)dcap);
    {
      Timed t("k_inf_scan");
      g.sv_ok = false;
      hipLaunchKernelGGL(k_inf_scan, dim3((uint32_t)chunks), dim3(INF_SCAN_THREADS), 0, g.stream, d_in, dbufs, 1u, (unsigned long long*)g.surv.p,
                         pd.surv_cap, counters, dfirst, (flags & ZES_F_LOOSE_CANDIDATES) ? 1u : 2u, (const uint8_t*)g.kraft.p);
    }
    if ((rc = launch_verify(d_in, dbufs, pd.surv_cap, counters, cnt, (flags & ZES_F_LOOSE_CANDIDATES) ? 1u : 0u, c, 32768))) return rc;
    {
      Timed t("k_inf_block_par");
      hipLaunchKernelGGL(two ? k_inf_block_par2 : k_inf_block_par, dim3(pd.bound), dim3(PAR_THREADS), 0, g.stream, d_in, d_out, dbufs, 1u,
                         (const uint32_t*)cnt, (const uint32_t*)g.cand_sorted.p, (const uint32_t*)nullptr, (ZesCandRes*)g.cres.p,
                         (unsigned long long*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)g.cand.p, (uint32_t*)g.cand_sorted.p, ZesParMirror{});
    }
    {
      Timed t("k_inf_chain");
      hipLaunchKernelGGL(k_inf_chain_range, dim3(1), dim3(256), 0, g.stream, dbufs, (const uint32_t*)cnt, (const uint32_t*)g.cand_sorted.p,
                         (const ZesCandRes*)g.cres.p, (ZesRes*)g.res.p, range_acc());
    }
    HIPCHK(hipMemcpyAsync(range_hc(slot), g.counters.p, 24, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipMemcpyAsync(range_hres(slot), g.res.p, sizeof(ZesRes) * 2, hipMemcpyDeviceToHost, g.stream));
  }
  HIPCHK(hipEventRecord(g.ev_rng[slot], g.stream));
  return ZES_OK;
}
int range_finish(int slot, const RangePend& pd, RangeRes* rr) {
  *rr = RangeRes();
  HIPCHK(hipEventSynchronize(g.ev_rng[slot]));
  if (pd.skip) return ZES_OK;
  auto nothing_here = [&]() {  // a range without a block start (a piece in the middle of one block): nothing to decode, and that is an answer
    if (pd.exact) return ZES_OK;
    rr->handled = true;
    rr->first_bit = rr->end_bit = pd.lo_bit;
    return ZES_OK;
  };
  if (pd.nothing) return nothing_here();
  const uint32_t* hc = range_hc(slot);
  const ZesRes* hres = range_hres(slot);
  const uint32_t nsurv = hc[0], ncand = hc[4];
  if (nsurv > pd.surv_cap || ncand > pd.cand_cap) return ZES_OK;
  if (nsurv == 0 || ncand == 0) return nothing_here();
  if (ncand > pd.bound) return ZES_OK;  // (more candidates than work items were launched: not decoded this way)
  if (hres[0].status != 0) return ZES_OK;
  rr->handled = true;
  rr->out_len = hres[0].out_len;
  rr->nblocks = hres[0].aux & 0x7FFFFFFFu;
  rr->final_block = (hres[0].aux >> 31) != 0;
  rr->end_bit = hres[1].out_len;
  rr->first_bit = hres[1].aux;
  return ZES_OK;
}

// T2 for the buffers the block-parallel tier did not settle: segment-parallel decode of any valid stream
// (blocks of every type, 32 KiB history across blocks).  The candidate search runs buffer by buffer; the segment
// decode — nearly all of the time — and the window pass take all buffers of a group in one launch.  A buffer whose
// stream is not a clean chain of blocks keeps tier 0: the serial tiers then reproduce the reference's result.
#ifndef ZES_SEG_MIN_C
#define ZES_SEG_MIN_C 4096
#endif
constexpr uint64_t SEG_MIN_C = ZES_SEG_MIN_C;  // shorter streams go straight to the serial wavefront (round 3: 32768 -> 4096: 29 KB of zlib stream can be 4 MiB of periodic data — 12.5 ms by the serial wavefront, 2.8 ms here)
constexpr size_t SERIAL_BATCH_MIN_JOBS = 16;        // this many left-over streams of a call: one serial wavefront each, side by side
constexpr uint64_t SERIAL_BATCH_MAX_C = 8ull << 10;  // (round 3: 128 KiB -> 8 KiB)  // (longer ones go to the segment-parallel tier: its block decoder is ~15 times a lone wave)
constexpr uint64_t SEG_PIECES_MIN_C = 48ull << 20;  // streams from this size on go through the tier in pieces of 32 MiB (inflate_segments_pieces)
constexpr uint32_t SEG_GROUP_BUFS = 512;     // (round 3: 64 -> 512: 2048 x 64 KiB of zlib text 7.4 -> 3.8 ms with single-block streams taken by the block decoder)     // buffers whose candidates are searched before the first read-back
constexpr uint32_t SEG_GROUP_WORK = 8192;    // work items per segment launch (each owns a 64 KiB map)

// One group: buffers ids[0..nb) with their sorted candidate lists at cand_sorted + cbase[k], ncand[k] entries.
int inflate_segments_run(const uint8_t* d_in, uint8_t* d_out, InfJob* jobs, const uint32_t* ids, const uint32_t* cbase, const uint32_t* ncand,
                         uint32_t nb, uint32_t* dscratch /* 2 * nb + 1 words: chain segments not in the store, failure flags, far-match counter */) {
  int rc;
  ZesSegJob* hj = (ZesSegJob*)((uint8_t*)g.pinned + PIN_UP);
  uint32_t work = 0;
  uint64_t csum = 0;
  for (uint32_t k = 0; k < nb; k++) csum += jobs[ids[k]].c + 64;
  // symbol store: `ratio` 16-bit symbols per compressed byte (a segment that inflates further is decoded twice);
  // as many as a 2 GiB store holds, up to DEFLATE's own limit of 1032 bytes per compressed byte
  uint32_t ratio = (uint32_t)std::min<uint64_t>(1032, (2ull << 30) / (2 * csum)) & ~1u;
  if (ratio < 4) ratio = 0;
  // behind the shares: a common area of a quarter of their size (at least 64 MiB), handed out by need to blocks whose
  // share is too small for them (a false candidate inside the block has cut it short, or the block inflates further)
  const uint64_t share_syms = (uint64_t)csum * ratio;
  const uint64_t bump_syms = ratio ? std::max<uint64_t>(share_syms / 4, 32ull << 20) & ~7ull : 0;
  if (ratio && ensure(g.sym16, (size_t)(share_syms + bump_syms) * 2 + 64)) ratio = 0;  // no memory for it: two decodes
  uint64_t sym_base = 0;
  for (uint32_t k = 0; k < nb; k++) {
    const InfJob& j = jobs[ids[k]];
    hj[k].in_off = j.in_off;
    hj[k].c = j.c;
    hj[k].sym_base = sym_base;
    hj[k].cand_base = cbase[k];
    hj[k].ncand = ncand[k];
    hj[k].work_first = work;
    hj[k].nseg = 0;
    hj[k].start0 = j.start0;
    hj[k].flags = j.partial ? ZES_SEG_PARTIAL : 0u;
    work += ncand[k] + 1;
    sym_base += (j.c + 64) * ratio / 2;
  }
  if ((rc = ensure(g.segjobs, sizeof(ZesSegJob) * nb))) return rc;
  if ((rc = ensure(g.sres, sizeof(ZesSegRes) * work))) return rc;
  if ((rc = ensure(g.maps, (size_t)work * ZES_WINDOW * 2))) return rc;
  if ((rc = ensure(g.seglist, (size_t)work * 4))) return rc;
  if ((rc = ensure(g.segprefix, (size_t)work * 8))) return rc;
  if ((rc = ensure(g.segorder, (size_t)work * 4))) return rc;
  if ((rc = ensure(g.symoff, (size_t)work * 8 + 8))) return rc;  // per work item: where its symbols are; behind them: the common area's fill
  if ((rc = ensure(g.res, sizeof(ZesRes) * nb * 2))) return rc;  // (second half: where a chain ended, k_inf_seg_chain)
  HIPCHK(hipMemcpyAsync(g.segjobs.p, hj, sizeof(ZesSegJob) * nb, hipMemcpyHostToDevice, g.stream));
  HIPCHK(hipMemsetAsync(dscratch, 0, (size_t)nb * 8 + 4, g.stream));
  uint32_t* novf_d = dscratch;
  uint32_t* fail_d = dscratch + nb;
  const uint32_t* cs = (const uint32_t*)g.cand_sorted.p;
  hipLaunchKernelGGL(k_inf_seg_order, dim3(nb), dim3(1024), 0, g.stream, (const ZesSegJob*)g.segjobs.p, cs, (uint32_t*)g.segorder.p);
  // The decoders run with the short marker ring first (three per CU instead of two); a match that reaches behind the
  // ring takes its symbols from the symbol store, so a segment that has outgrown its share of the store and then meets
  // such a match cannot go on: far_d counts those, and the whole group runs again with the full ring (rare: streams
  // that inflate by more than the store's symbols per compressed byte).
  uint32_t* far_d = dscratch + 2 * nb;
  // Every work item first goes to the block decoder of the block-parallel tier in its any-encoder form
  // (k_inf_seg_block_par: a workgroup per block, 1024 lanes decoding 1024 bit segments of it) — a wave that decodes a
  // block token by token gets through ~13 MB/s.  What it declines (an item whose block is stored or fixed, is followed
  // by a block that is not on the list, or is longer than 128 KiB) is listed, and the wave decoder runs for the list.
  const bool blockpar = ratio != 0 && !getenv("ZES_NO_SEG_PAR");
  uint32_t* fail_list = nullptr;
  if (blockpar) {
    if ((rc = ensure(g.segfail, ((size_t)work + 1) * 4))) return rc;
    fail_list = (uint32_t*)g.segfail.p;
    HIPCHK(hipMemsetAsync(fail_list, 0, 4, g.stream));
    HIPCHK(hipMemsetAsync((uint64_t*)g.symoff.p + work, 0, 8, g.stream));
    unsigned long long* pdbg = nullptr;
    if (getenv("ZES_DEBUG_PHASES")) {
      if ((rc = ensure(g.dbg, (size_t)work * ZES_PAR_DBG_ROW * 8))) return rc;
      HIPCHK(hipMemsetAsync(g.dbg.p, 0, (size_t)work * ZES_PAR_DBG_ROW * 8, g.stream));
      pdbg = (unsigned long long*)g.dbg.p;
    }
    {
      Timed t("k_inf_seg_block_par");
      hipLaunchKernelGGL(k_inf_seg_block_par, dim3(work), dim3(PAR_THREADS), 0, g.stream, d_in, (const ZesSegJob*)g.segjobs.p, nb, cs, (ZesSegRes*)g.sres.p,
                         (uint32_t*)g.maps.p, (uint32_t*)g.sym16.p, ratio, fail_list, (unsigned long long*)((uint64_t*)g.symoff.p + work),
                         (uint64_t)((share_syms / 2 + 3) & ~3ull), bump_syms, (uint64_t*)g.symoff.p, pdbg);
    }
    if (pdbg) {
      HIPCHK(hipStreamSynchronize(g.stream));
      if ((rc = print_par_phases(pdbg, work))) return rc;
    }
  }
  auto dump_items = [&](const char* tag) {  // ZES_T2_DBG: what every work item has come to so far
    if (!getenv("ZES_T2_DBG")) return;
    (void)hipStreamSynchronize(g.stream);
    std::vector<ZesSegRes> sr(work);
    std::vector<uint32_t> fl(work + 1, 0);
    (void)hipMemcpy(sr.data(), g.sres.p, sr.size() * sizeof(ZesSegRes), hipMemcpyDeviceToHost);
    if (fail_list) (void)hipMemcpy(fl.data(), fail_list, fl.size() * 4, hipMemcpyDeviceToHost);
    fprintf(stderr, "zes T2 items %s (fail list: %u:", tag, fl[0]);
    for (uint32_t i = 0; i < fl[0] && i < 16; i++) fprintf(stderr, " %u", fl[1 + i]);
    fprintf(stderr, ")\n");
    for (size_t w = 0; w < sr.size(); w++)
      fprintf(stderr, "   item %zu: end_bit %llu out_len %llu flags %u next %u\n", w, (unsigned long long)sr[w].end_bit,
              (unsigned long long)sr[w].out_len, sr[w].flags, sr[w].next);
  };
  dump_items("after the block decoder");
  uint32_t* hs = (uint32_t*)g.pinned;  // [0, nb): not-in-store counts, later failure flags; [nb]: declined items
  ZesRes* hres = (ZesRes*)((uint8_t*)g.pinned + 128 * 1024);
  // the chain of every buffer of the group (work item 0 -> the item that starts where it ended -> ... -> the final block)
  auto run_chains = [&]() -> int {
    HIPCHK(hipMemsetAsync(novf_d, 0, (size_t)nb * 4, g.stream));
    {
      Timed t("k_inf_seg_chain");  // (one launch: a workgroup per buffer of the group)
      hipLaunchKernelGGL(k_inf_seg_chain, dim3(nb), dim3(256), 0, g.stream, (const ZesSegJob*)g.segjobs.p, (const ZesSegRes*)g.sres.p,
                         (uint32_t*)g.seglist.p, (uint64_t*)g.segprefix.p, (ZesRes*)g.res.p, novf_d);
    }
    HIPCHK(hipMemcpyAsync(hs, novf_d, (size_t)nb * 4, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipMemcpyAsync(hres, g.res.p, sizeof(ZesRes) * nb * 2, hipMemcpyDeviceToHost, g.stream));
    if (blockpar) HIPCHK(hipMemcpyAsync(hs + nb, fail_list, 4, hipMemcpyDeviceToHost, g.stream));
    host_lap("(host work since)");
    HIPCHK(hipStreamSynchronize(g.stream));  // (the job table upload has completed too: hj may be rewritten)
    host_lap("T2: decode / chains");
    return ZES_OK;
  };
  // Round 3: the chains are tried on what the block decoder left BEFORE the wave decoder gets the declined items.  An
  // item that starts on a false candidate is declined (its "block" is garbage) and nobody's chain leads to it — but
  // the lone wave that decodes it on may take milliseconds to find that out (256 x 1 MiB of zlib text: 6.9 of 32 ms).
  // Only when a chain does run into a declined item (a stored or fixed block, a block behind an unlisted start) do the
  // declined items go to the wave decoder, and the chains are followed again.
  bool chained = false;
  if (blockpar) {
    if ((rc = run_chains())) return rc;
    chained = true;
    const uint32_t ndecl = hs[nb];
    if (getenv("ZES_DEBUG")) fprintf(stderr, "zes T2: %u work items, %u declined by the block decoder\n", work, ndecl);
    // Chains that stand in front of an undecoded item: exactly those items go to the wave decoder (a zlib stream's
    // last block is often a fixed one: one small item per stream, where the declined list of 256 streams of 1 MiB also
    // held ~200 false candidates that cost the lone waves 6.8 ms), and the chains are followed again; a few rounds,
    // then — streams with many stored or fixed blocks — everything that was declined.
    std::vector<uint32_t> last_stuck(nb, 0xFFFFFFFFu);
    // the item a buffer's chain stands in front of (status 1: no chain; 3: the chain of a piece, as far as it got)
    auto stuck_of = [&](uint32_t k) -> uint32_t {
      if (hres[k].status == 1 && hres[k].out_len != 0) return (uint32_t)(hres[k].out_len - 1);
      if (hres[k].status == 3 && hres[nb + k].aux != 0) return hres[nb + k].aux - 1u;
      return 0xFFFFFFFFu;
    };
    // (Not when an eighth of all items were declined: a stream of blocks of a few hundred bytes — zlib with memLevel 1 —
    // is thinned to 2048 items of a dozen blocks each, every one of them handed over behind its first block: four
    // rounds of one lone wave each were 16 of that stream's 31 ms before the launch that takes them all.)
    const bool many = ndecl >= 16 && (uint64_t)ndecl * 8 >= work;
    for (int round = 0; ndecl && !many && round < 4; round++) {
      std::vector<uint32_t> items;
      for (uint32_t k = 0; k < nb; k++) {
        const uint32_t w = stuck_of(k);
        if (w == 0xFFFFFFFFu || w == last_stuck[k]) continue;  // (the same item again: it has been to the wave decoder — the stream is not for this tier)
        // a piece of a longer stream whose chain got through half the piece: what it stands in front of is, as a rule,
        // the block the piece's end cuts — the next piece starts there
        if (hres[k].status == 3 && hres[nb + k].out_len >= jobs[ids[k]].c * 4) continue;
        last_stuck[k] = w;
        items.push_back(hj[k].work_first + w);
      }
      if (items.empty()) break;
      if ((rc = ensure(g.seglive, ((size_t)nb + 1) * 4))) return rc;
      uint32_t* hl = (uint32_t*)((uint8_t*)g.pinned + 768 * 1024);  // (upload area, behind the job table)
      hl[0] = (uint32_t)items.size();
      memcpy(hl + 1, items.data(), items.size() * 4);
      HIPCHK(hipMemcpyAsync(g.seglive.p, hl, (items.size() + 1) * 4, hipMemcpyHostToDevice, g.stream));
      {
        Timed t("k_inf_seg_scan");
        hipLaunchKernelGGL(k_inf_seg_scan_short, dim3((uint32_t)items.size()), dim3(64), 0, g.stream, d_in, (const ZesSegJob*)g.segjobs.p, nb, cs,
                           (ZesSegRes*)g.sres.p, (uint32_t*)g.maps.p, (uint32_t*)g.sym16.p, ratio, (const uint32_t*)g.seglive.p + 1, far_d,
                           (const uint32_t*)g.seglive.p, (uint64_t*)g.symoff.p);
      }
      uint32_t* hf = (uint32_t*)((uint8_t*)g.pinned + 192 * 1024);
      HIPCHK(hipMemcpyAsync(hf, far_d, 4, hipMemcpyDeviceToHost, g.stream));
      if ((rc = run_chains())) return rc;
      if (hf[0] != 0) {  // a far match behind the short ring: the full-ring pass below decides
        chained = false;
        break;
      }
    }
    if (chained && ndecl)
      for (uint32_t k = 0; k < nb; k++) {
        const uint32_t w = stuck_of(k);
        if (w == 0xFFFFFFFFu || w == last_stuck[k]) continue;
        // rounds used up with a chain still in front of an undecoded item (a stream whose blocks mostly follow blocks
        // that are not on the thinned list): everything that was declined goes to the wave decoder in one launch —
        // unless it is a piece's chain that got through half the piece (see above)
        if (hres[k].status == 3 && hres[nb + k].out_len >= jobs[ids[k]].c * 4) continue;
        chained = false;
      }
  }
  if (!chained) {
    {
      Timed t("k_inf_seg_scan");
      hipLaunchKernelGGL(k_inf_seg_scan_short, dim3(work), dim3(64), 0, g.stream, d_in, (const ZesSegJob*)g.segjobs.p, nb, cs, (ZesSegRes*)g.sres.p,
                         (uint32_t*)g.maps.p, (uint32_t*)g.sym16.p, ratio, blockpar ? (const uint32_t*)fail_list + 1 : (const uint32_t*)g.segorder.p, far_d,
                         (const uint32_t*)fail_list, (uint64_t*)g.symoff.p);
    }
    {
      uint32_t* hf = (uint32_t*)g.pinned;
      HIPCHK(hipMemcpyAsync(hf, far_d, 4, hipMemcpyDeviceToHost, g.stream));
      HIPCHK(hipStreamSynchronize(g.stream));
      if (hf[0] != 0) {
        Timed t("k_inf_seg_scan");
        hipLaunchKernelGGL(k_inf_seg_scan, dim3(work), dim3(64), 0, g.stream, d_in, (const ZesSegJob*)g.segjobs.p, nb, cs, (ZesSegRes*)g.sres.p,
                           (uint32_t*)g.maps.p, (uint32_t*)g.sym16.p, ratio, (const uint32_t*)g.segorder.p, far_d, (const uint32_t*)nullptr,
                           (uint64_t*)g.symoff.p);
      }
    }
    if ((rc = run_chains())) return rc;
  }
  std::vector<uint32_t> novf(hs, hs + nb);
  std::vector<ZesRes> hr(hres, hres + 2 * nb);
  std::vector<char> go(nb, 0);
  bool any = false;
  for (uint32_t k = 0; k < nb; k++) {
    InfJob& j = jobs[ids[k]];
    if (getenv("ZES_DEBUG"))
      fprintf(stderr, "zes T2: c=%llu candidates=%u chain status=%d segments=%u (%u decoded twice) out_len=%llu\n", (unsigned long long)j.c,
              ncand[k], hr[k].status, hr[k].aux, novf[k], (unsigned long long)hr[k].out_len);
    if (getenv("ZES_T2_DBG")) {  // what every work item came to
      std::vector<ZesSegRes> sr(ncand[k] + 1);
      HIPCHK(hipMemcpy(sr.data(), (const ZesSegRes*)g.sres.p + hj[k].work_first, sr.size() * sizeof(ZesSegRes), hipMemcpyDeviceToHost));
      for (size_t w = 0; w < sr.size(); w++)
        fprintf(stderr, "   item %zu: end_bit %llu out_len %llu flags %u next %u\n", w, (unsigned long long)sr[w].end_bit,
                (unsigned long long)sr[w].out_len, sr[w].flags, sr[w].next);
    }
    if (hr[k].status == 3 && j.partial) hr[k].status = 0;  // a piece's chain, as far as it got
    if (hr[k].status != 0 || hr[k].aux == 0) continue;
    j.end_bit = hr[nb + k].out_len;
    j.final_seen = hr[nb + k].status != 0;
    if (hr[k].out_len > j.cap) {  // the caller learns the size without the output passes
      j.tier = 2;
      j.out_len = hr[k].out_len;
      j.status = ZES_E_NOSPACE;
      continue;
    }
    go[k] = 1;
    any = true;
    hj[k].nseg = hr[k].aux;
  }
  if (!any) return ZES_OK;
  if ((rc = ensure(g.wins, (size_t)work * ZES_WINDOW))) return rc;
  HIPCHK(hipMemcpyAsync(g.segjobs.p, hj, sizeof(ZesSegJob) * nb, hipMemcpyHostToDevice, g.stream));
  // where every buffer's bytes go (and how much output exists in front: a later piece of a long stream)
  if ((rc = ensure(g.segouts, sizeof(ZesSegOut) * nb))) return rc;
  ZesSegOut* ho = (ZesSegOut*)((uint8_t*)g.pinned + 832 * 1024);  // (upload area)
  uint32_t max_tr = 0, min_tr = 0xFFFFFFFFu;
  for (uint32_t k = 0; k < nb; k++) {
    const InfJob& j = jobs[ids[k]];
    ho[k].out_off = j.out_off;
    ho[k].cap = j.cap;
    ho[k].nseg = (go[k] && novf[k] < hr[k].aux) ? hr[k].aux : 0u;
    ho[k].hist = j.hist;
    if (ho[k].nseg) {
      max_tr = std::max(max_tr, ho[k].nseg);
      min_tr = std::min(min_tr, ho[k].nseg);
    }
  }
  HIPCHK(hipMemcpyAsync(g.segouts.p, ho, sizeof(ZesSegOut) * nb, hipMemcpyHostToDevice, g.stream));
  {
    // windows: groups of maps composed in parallel, the groups chained, every window finished in parallel
    uint32_t max_nseg = 0;
    for (uint32_t k = 0; k < nb; k++) max_nseg = std::max(max_nseg, hj[k].nseg);
    const uint32_t max_groups = (max_nseg + SEGWIN_GROUP - 1) / SEGWIN_GROUP;
    if ((rc = ensure(g.pw16, (size_t)work * ZES_WINDOW * 2))) return rc;
    if ((rc = ensure(g.gwins, ((size_t)work / SEGWIN_GROUP + nb + 1) * ZES_WINDOW))) return rc;  // (work_first / group) + buffer index + group
    Timed t("k_inf_seg_windows");
    hipLaunchKernelGGL(k_inf_seg_win_group, dim3(max_groups, nb), dim3(1024), 0, g.stream, (const uint32_t*)g.maps.p,
                       (const uint32_t*)g.seglist.p, (const ZesSegJob*)g.segjobs.p, (uint32_t*)g.pw16.p);
    hipLaunchKernelGGL(k_inf_seg_win_top, dim3(nb), dim3(1024), 0, g.stream, (const uint32_t*)g.pw16.p, (const ZesSegJob*)g.segjobs.p,
                       (uint8_t*)g.gwins.p, (const uint8_t*)d_out, (const ZesSegOut*)g.segouts.p);
    hipLaunchKernelGGL(k_inf_seg_win_fin, dim3(max_nseg, nb), dim3(1024), 0, g.stream, (const uint32_t*)g.pw16.p,
                       (const ZesSegJob*)g.segjobs.p, (const uint8_t*)g.gwins.p, (uint8_t*)g.wins.p);
  }
  {
    // symbols -> bytes: one launch over (segments, workgroups per segment, buffers)
    if (max_tr) {
      Timed t("k_inf_seg_translate");
      // few long segments: split each over several workgroups (by the buffer with the fewest)
      const uint32_t ny = std::max(1u, std::min(16u, 2048u / std::max(1u, min_tr * std::min(nb, 8u))));
      hipLaunchKernelGGL(k_inf_seg_translate, dim3(max_tr, ny, nb), dim3(256), 0, g.stream, d_out, (const ZesSegJob*)g.segjobs.p,
                         (const ZesSegOut*)g.segouts.p, cs, (const ZesSegRes*)g.sres.p, (const uint32_t*)g.seglist.p,
                         (const uint64_t*)g.segprefix.p, (const uint8_t*)g.wins.p, (const uint32_t*)g.sym16.p, (const uint64_t*)g.symoff.p, fail_d);
    }
  }
  for (uint32_t k = 0; k < nb; k++) {
    if (!go[k]) continue;
    const InfJob& j = jobs[ids[k]];
    const uint32_t nseg = hr[k].aux, wf = hj[k].work_first;
    if (novf[k] > 0) {
      Timed t("k_inf_seg_decode");
      hipLaunchKernelGGL(k_inf_seg_decode, dim3(nseg), dim3(64), 0, g.stream, d_in, j.in_off, j.c, d_out, j.out_off, j.cap, cs + cbase[k],
                         (const ZesSegRes*)g.sres.p + wf, (const uint32_t*)g.seglist.p + wf, (const uint64_t*)g.segprefix.p + wf,
                         (const uint8_t*)g.wins.p + (size_t)wf * ZES_WINDOW, fail_d + k, novf[k] < nseg ? 1u : 0u, j.start0, j.hist);
    }
  }
  HIPCHK(hipMemcpyAsync(hs, fail_d, (size_t)nb * 4, hipMemcpyDeviceToHost, g.stream));
  host_lap("(host work since)");
  HIPCHK(hipStreamSynchronize(g.stream));
  host_lap("T2: windows + translate");
  for (uint32_t k = 0; k < nb; k++) {
    if (!go[k] || hs[k] != 0) continue;  // (a match behind the first byte of the stream: the serial tiers decide)
    InfJob& j = jobs[ids[k]];
    j.tier = 2;
    j.out_len = hr[k].out_len;
    j.status = ZES_OK;
  }
  return ZES_OK;
}

int inflate_segments(const uint8_t* d_in, uint8_t* d_out, InfJob* jobs, const std::vector<uint32_t>& all) {
  int rc;
  for (size_t g0 = 0; g0 < all.size(); g0 += SEG_GROUP_BUFS) {
    const uint32_t nb = (uint32_t)std::min<size_t>(SEG_GROUP_BUFS, all.size() - g0);
    const uint32_t* ids = all.data() + g0;
    // ---- candidates, buffer by buffer (each search has the chip to itself) ----
    std::vector<uint32_t> cbase(nb), ccap(nb);
    uint64_t cands = 0, max_c = 0;
    for (uint32_t k = 0; k < nb; k++) {
      const InfJob& j = jobs[ids[k]];
      cbase[k] = (uint32_t)cands;
      ccap[k] = (uint32_t)(j.c / 64 + 64);
      cands += ccap[k];
      max_c = std::max(max_c, j.c);
    }
    const uint32_t surv_cap = (uint32_t)std::min<uint64_t>(max_c / 4 + 1024ull, 1ull << 30);
    const size_t cnt_words = 4 + (size_t)nb + 2 * (size_t)nb + 4 + 4 + ((size_t)nb + 3) / 4;  // scan/verify scratch, counts, run scratch (+ far-match counter), first-byte sink (a byte per buffer)
    if ((rc = ensure(g.ibufs, sizeof(ZesInfBuf) * 2))) return rc;
    if ((rc = ensure(g.surv, (size_t)surv_cap * 8))) return rc;
    if ((rc = ensure(g.cand, (size_t)cands * 4))) return rc;
    if ((rc = ensure(g.cand_sorted, (size_t)cands * 4))) return rc;
    if ((rc = ensure(g.counters, cnt_words * 4))) return rc;
    if ((rc = ensure(g.mvlist, SEG_BUCKETS * 4))) return rc;
    uint32_t* counters = (uint32_t*)g.counters.p;
    uint32_t* cnt = counters + 4;
    uint32_t* dscratch = cnt + nb;
    uint8_t* sink = (uint8_t*)(dscratch + 2 * nb + 4);
    HIPCHK(hipMemsetAsync(counters, 0, cnt_words * 4, g.stream));
    const ZesInfBuf* dbufs = (const ZesInfBuf*)g.ibufs.p;
    // Several buffers (a batch of another encoder's streams): one scan and one header test over all of them, the lists
    // sorted by one launch; only a buffer with more candidates than a segment run takes (tiny blocks) is thinned, by
    // itself, afterwards.  (256 streams of 1 MiB: the searches one after the other were 36 ms of launches.)
    std::vector<char> searched(nb, 0);
    if (nb > 1) {
      ZesInfBuf* hb = (ZesInfBuf*)((uint8_t*)g.pinned + PIN_UP);
      memset(hb, 0, sizeof(ZesInfBuf) * (nb + 1));
      uint64_t chunks = 0, total_c = 0;
      for (uint32_t k = 0; k < nb; k++) {
        const InfJob& j = jobs[ids[k]];
        hb[k].in_off = j.in_off;
        hb[k].c = j.c;
        hb[k].out_off = j.out_off;
        hb[k].cap = j.cap;
        hb[k].first_chunk = (uint32_t)chunks;
        hb[k].cand_base = cbase[k];
        hb[k].cand_cap = ccap[k];
        hb[k].start_rel = j.start0 - 16u;  // (a piece of a longer stream: nothing in front of its first block is searched)
        chunks += (j.c + INF_SCAN_BYTES - 1) / INF_SCAN_BYTES;
        total_c += j.c;
      }
      hb[nb].first_chunk = (uint32_t)chunks;
      const uint32_t surv_all = (uint32_t)std::min<uint64_t>(total_c / 4 + 1024ull * nb, 1ull << 30);
      if ((rc = ensure(g.ibufs, sizeof(ZesInfBuf) * (nb + 1)))) return rc;
      if ((rc = ensure(g.surv, (size_t)surv_all * 8))) return rc;
      dbufs = (const ZesInfBuf*)g.ibufs.p;
      HIPCHK(hipMemcpyAsync(g.ibufs.p, hb, sizeof(ZesInfBuf) * (nb + 1), hipMemcpyHostToDevice, g.stream));
      {
        Timed t("k_inf_scan");
        g.sv_ok = false;
        hipLaunchKernelGGL(k_inf_scan, dim3((uint32_t)chunks), dim3(INF_SCAN_THREADS), 0, g.stream, d_in, dbufs, nb, (unsigned long long*)g.surv.p,
                           surv_all, counters, sink, 0u, (const uint8_t*)g.kraft.p);
      }
      if ((rc = launch_verify(d_in, dbufs, surv_all, counters, cnt, 1u, total_c, 16384))) return rc;
      {
        Timed t("k_inf_ranksort");
        hipLaunchKernelGGL(k_inf_ranksort, dim3(nb), dim3(1024), 0, g.stream, dbufs, (const uint32_t*)cnt, (const uint32_t*)g.cand.p,
                           (uint32_t*)g.cand_sorted.p, SEG_BUCKETS);
      }
      uint32_t* hc0 = (uint32_t*)g.pinned;
      HIPCHK(hipMemcpyAsync(hc0, cnt, (size_t)nb * 4, hipMemcpyDeviceToHost, g.stream));
      HIPCHK(hipStreamSynchronize(g.stream));
      for (uint32_t k = 0; k < nb; k++) searched[k] = hc0[k] <= SEG_BUCKETS;  // (the others: thinned below, from a search of their own)
      bool redo = false;
      for (uint32_t k = 0; k < nb; k++) redo = redo || !searched[k];
      if (redo) {
        if ((rc = ensure(g.ibufs2, sizeof(ZesInfBuf) * 2))) return rc;
        dbufs = (const ZesInfBuf*)g.ibufs2.p;
      }
    }
    for (uint32_t k = 0; k < nb; k++) {
      if (searched[k]) continue;
      HIPCHK(hipMemsetAsync(cnt + k, 0, 4, g.stream));
      const InfJob& j = jobs[ids[k]];
      ZesInfBuf b0, b1;
      memset(&b0, 0, sizeof b0);
      memset(&b1, 0, sizeof b1);
      b0.in_off = j.in_off;
      b0.c = j.c;
      b0.out_off = j.out_off;
      b0.cap = j.cap;
      b0.cand_base = cbase[k];
      b0.cand_cap = ccap[k];
      b0.start_rel = j.start0 - 16u;
      const uint32_t chunks = (uint32_t)((j.c + INF_SCAN_BYTES - 1) / INF_SCAN_BYTES);
      b1.first_chunk = chunks;
      hipLaunchKernelGGL(k_inf_set_table1, dim3(1), dim3(64), 0, g.stream, b0, b1, const_cast<ZesInfBuf*>(dbufs), counters, 4u);
      // The block-parallel tier has just searched this very stream and declined it (another encoder's): its scan's
      // survivors are still in g.surv — the scan applies the same tests for both tiers, the reference's own rules are the
      // verify kernels' — so only their count goes back into place (0.06 of the 1.65 ms of 64 MiB of zlib text).
      const bool reuse = nb == 1 && g.sv_ok && g.sv_list == g.surv.p && g.sv_din == d_in && g.sv_in_off == j.in_off && g.sv_c == j.c && j.start0 == 16u && g.sv_n <= surv_cap;
      g.sv_ok = false;
      if (reuse) {
        uint32_t* hv = (uint32_t*)((uint8_t*)g.pinned + 196 * 1024);
        hv[0] = g.sv_n;
        HIPCHK(hipMemcpyAsync(counters, hv, 4, hipMemcpyHostToDevice, g.stream));
      } else {
        Timed t("k_inf_scan");
        // (the BFINAL rule of the scan holds for every encoder's streams: it stays on; only the verify rules are the reference's own)
        g.sv_ok = false;
        hipLaunchKernelGGL(k_inf_scan, dim3(chunks), dim3(INF_SCAN_THREADS), 0, g.stream, d_in, dbufs, 1u, (unsigned long long*)g.surv.p,
                           surv_cap, counters, sink, 0u, (const uint8_t*)g.kraft.p);
      }
      // other encoders do not follow the reference's run-length rules for code lengths: loose candidates
      if ((rc = launch_verify(d_in, dbufs, surv_cap, counters, cnt + k, 1u, j.c, 16384))) return rc;
      {
        // one candidate per bucket of the stream, in order (at most SEG_BUCKETS segments whatever the block size)
        Timed t("k_inf_cand_thin");
        const uint32_t bucket_bits = (uint32_t)((j.c * 8 + SEG_BUCKETS - 1) / SEG_BUCKETS);
        HIPCHK(hipMemsetAsync(g.mvlist.p, 0xFF, SEG_BUCKETS * 4, g.stream));
        hipLaunchKernelGGL(k_inf_cand_bucket, dim3((uint32_t)std::min<uint64_t>(ccap[k] / 256 + 1, 1024)), dim3(256), 0, g.stream,
                           (const uint32_t*)g.cand.p + cbase[k], (const uint32_t*)(cnt + k), ccap[k], bucket_bits, (uint32_t*)g.mvlist.p);
        hipLaunchKernelGGL(k_inf_cand_compact, dim3(1), dim3(1024), 0, g.stream, (const uint32_t*)g.mvlist.p,
                           (uint32_t*)g.cand_sorted.p + cbase[k], cnt + k);
      }
    }
    uint32_t* hc = (uint32_t*)g.pinned;
    HIPCHK(hipMemcpyAsync(hc, cnt, (size_t)nb * 4, hipMemcpyDeviceToHost, g.stream));
    host_lap("(host work since)");
    HIPCHK(hipStreamSynchronize(g.stream));
    host_lap("T2: search (verify, thinning)");
    std::vector<uint32_t> nc(hc, hc + nb);
    if (getenv("ZES_T2_DBG")) {  // the candidate lists, for a comparison with a map of the stream (tools/gpu_t2_candidates.py)
      for (uint32_t k = 0; k < nb; k++) {
        std::vector<uint32_t> hcand(std::min<uint32_t>(nc[k], SEG_BUCKETS));
        if (!hcand.empty()) HIPCHK(hipMemcpy(hcand.data(), (const uint32_t*)g.cand_sorted.p + cbase[k], hcand.size() * 4, hipMemcpyDeviceToHost));
        fprintf(stderr, "zes T2 candidates buf %u (%zu):", k, hcand.size());
        for (uint32_t v : hcand) fprintf(stderr, " %u", v + 16u);
        fprintf(stderr, "\n");
      }
    }
    // ---- segment runs: as many buffers as fit the work-item budget at a time ----
    std::vector<uint32_t> rid, rbase, rn;
    uint32_t work = 0;
    auto flush = [&]() -> int {
      if (rid.empty()) return ZES_OK;
      const int r = inflate_segments_run(d_in, d_out, jobs, rid.data(), rbase.data(), rn.data(), (uint32_t)rid.size(), dscratch);
      rid.clear();
      rbase.clear();
      rn.clear();
      work = 0;
      return r;
    };
    for (uint32_t k = 0; k < nb; k++) {
      if (nc[k] > SEG_BUCKETS) continue;  // (a poisoned count; a stream without a second block start is still one block for the block decoder)
      if (work + nc[k] + 1 > SEG_GROUP_WORK && (rc = flush())) return rc;
      rid.push_back(ids[k]);
      rbase.push_back(cbase[k]);
      rn.push_back(nc[k]);
      work += nc[k] + 1;
    }
    if ((rc = flush())) return rc;
  }
  return ZES_OK;
}

// A stream of stored blocks only (tier 2 as well: parallel, any encoder).  Leaves j.tier at 0 if it is anything else.
constexpr uint64_t STORED_MIN_C = 65536;  // below this the serial wavefront is as quick
// A stream too long for this tier's 32-bit bit positions (512 MiB of compressed data and more) goes through it piece by
// piece: a piece starts at the block behind the last one decoded (work item 0 at that bit, the candidate search from
// there), its chain is accepted as far as the piece holds whole blocks, and the 32 KiB in front of its first segment
// are the output so far.  (src/inflate.ts:16-40 has no size limit; reference-made streams of this size take
// inflate_pieces, the block-parallel tier's form of the same.)  ZES_SEG_PIECE_MB: piece size, for tests.
int inflate_segments_pieces(const uint8_t* d_in, uint8_t* d_out, InfJob& j) {
  static const uint64_t piece = [] {
    const char* e = getenv("ZES_SEG_PIECE_MB");
    const uint64_t mb = e ? strtoull(e, nullptr, 10) : 0;
    return mb ? (mb << 20) : (32ull << 20);  // (a piece of 32 MiB has about as many blocks as the tier takes work items: no thinning)
  }();
  int rc;
  uint64_t pos_bit = 16, out_base = 0;
  uint64_t piece_now = piece;  // (grows when a piece does not hold its first block whole: an encoder with very long blocks)
  bool nospace = false;  // the caller's room ran out: the pieces behind are only measured (the chain's lengths need no output)
  for (int guard = 0; guard < (1 << 20); guard++) {
    // the piece: from a 16-byte boundary at least two bytes in front of the block (work item 0 starts at bit >= 16 of it)
    const uint64_t pb = pos_bit >> 3;
    const uint64_t byte0 = pb >= 2 ? ((pb - 2) & ~15ull) : 0;
    std::vector<InfJob> sub(1);
    const uint64_t room = (nospace || j.cap <= out_base) ? 0 : j.cap - out_base;
    sub[0] = InfJob{j.in_off + byte0, std::min<uint64_t>(j.c - byte0, piece_now), j.out_off + std::min(out_base, j.cap), room, 0, ZES_OK, 0};
    sub[0].start0 = (uint32_t)(pos_bit - 8 * byte0);
    sub[0].hist = nospace ? 0u : (uint32_t)std::min<uint64_t>(out_base, ZES_WINDOW);
    sub[0].partial = byte0 + sub[0].c < j.c;  // (the last piece must end with the stream's final block)
    const std::vector<uint32_t> one(1, 0u);
    if ((rc = inflate_segments(d_in, d_out, sub.data(), one))) return rc;
    const bool stuck = sub[0].tier != 2 || (sub[0].status == ZES_OK && !sub[0].final_seen && 8 * byte0 + sub[0].end_bit <= pos_bit);
    if (stuck && sub[0].partial && piece_now < (448ull << 20)) {  // no whole block in this piece: a longer one, same start
      piece_now = std::min<uint64_t>(piece_now * 4, 448ull << 20);
      continue;
    }
    if (sub[0].tier != 2) return ZES_OK;  // not this way: the serial tiers decide
    piece_now = piece;
    if (sub[0].status == ZES_E_NOSPACE) nospace = true;
    else if (sub[0].status != ZES_OK) return ZES_OK;
    out_base += sub[0].out_len;
    if (sub[0].final_seen) {
      j.tier = 2;
      j.status = (nospace || out_base > j.cap) ? ZES_E_NOSPACE : ZES_OK;
      j.out_len = out_base;
      return ZES_OK;
    }
    const uint64_t next_bit = 8 * byte0 + sub[0].end_bit;
    if (!sub[0].partial || next_bit <= pos_bit) return ZES_OK;  // no final block where the data ends, or no progress
    pos_bit = next_bit;
  }
  return ZES_OK;
}

int inflate_stored(const uint8_t* d_in, uint8_t* d_out, InfJob& j) {
  int rc;
  const uint64_t cap_entries = std::min<uint64_t>(j.c / 5 + 1, 1ull << 22);
  if ((rc = ensure(g.res, sizeof(ZesRes)))) return rc;
  if ((rc = ensure(g.scratch, (size_t)cap_entries * sizeof(ZesStoredBlk)))) return rc;
  ZesRes hr;
  hr.status = 1;
  // up to 128 MiB: every byte position tested for a stored block's header, the chain from the first one marked by
  // pointer doubling (k_inf_stored_find / k_inf_stored_rank); what that cannot settle, and longer streams: the walk
  constexpr uint64_t STORED_PAR_MAX_C = 128ull << 20;
  constexpr uint32_t STORED_CHUNK_H = 16384, STORED_SLOTS_H = 8;
  if (j.c >= (2ull << 20) && j.c <= STORED_PAR_MAX_C && !getenv("ZES_NO_STORED_PAR")) {  // (a short stream's walk is quicker than two launches)
    const uint32_t nchunks = (uint32_t)((j.c + STORED_CHUNK_H - 1) / STORED_CHUNK_H);
    if ((rc = ensure(g.mvlist, (size_t)nchunks * (STORED_SLOTS_H + 1) * 4))) return rc;
    uint32_t* slots = (uint32_t*)g.mvlist.p;
    uint32_t* counts = slots + (size_t)nchunks * STORED_SLOTS_H;
    {
      Timed t("k_inf_stored_find");
      hipLaunchKernelGGL(k_inf_stored_find, dim3(nchunks), dim3(256), 0, g.stream, d_in, j.in_off, (uint32_t)j.c, slots, counts);
    }
    {
      Timed t("k_inf_stored_rank");
      hipLaunchKernelGGL(k_inf_stored_rank, dim3(1), dim3(1024), 0, g.stream, d_in, j.in_off, (uint32_t)j.c, (const uint32_t*)slots,
                         (const uint32_t*)counts, nchunks, cap_entries, (ZesStoredBlk*)g.scratch.p, (ZesRes*)g.res.p);
    }
    if ((rc = read_res(&hr))) return rc;
  }
  if (hr.status != 0) {
    Timed t("k_inf_stored_walk");
    hipLaunchKernelGGL(k_inf_stored_walk, dim3(1), dim3(64), 0, g.stream, d_in, j.in_off, j.c, cap_entries, (ZesStoredBlk*)g.scratch.p,
                       (ZesRes*)g.res.p);
    if ((rc = read_res(&hr))) return rc;
  }
  if (hr.status != 0) return ZES_OK;
  j.tier = 2;
  j.out_len = hr.out_len;
  j.status = hr.out_len > j.cap ? ZES_E_NOSPACE : ZES_OK;
  if (j.status == ZES_OK && hr.aux) {
    Timed t("k_inf_stored_copy");
    hipLaunchKernelGGL(k_inf_stored_copy, dim3(hr.aux), dim3(256), 0, g.stream, d_in, j.in_off, d_out, j.out_off, (const ZesStoredBlk*)g.scratch.p);
    HIPCHK(hipStreamSynchronize(g.stream));
  }
  return ZES_OK;
}

// T3 then T4 for one buffer the parallel tiers did not settle.
int inflate_slow(const uint8_t* d_in, uint8_t* d_out, InfJob& j) {
  int rc;
  if ((rc = ensure(g.res, sizeof(ZesRes)))) return rc;
  if ((rc = ensure(g.resume, 16))) return rc;
  ZesRes hr;
  bool have_resume = false;
  // T3: one wavefront, any valid stream
  if (j.c >= 3) {
#ifdef WD_PROFILE
    if ((rc = ensure(g.dbg, 64))) return rc;
    HIPCHK(hipMemsetAsync(g.dbg.p, 0, 64, g.stream));
    zes_wd_set_dbg((unsigned long long*)g.dbg.p);
#endif
    {
      ZesInfBuf b0;
      memset(&b0, 0, sizeof b0);
      b0.in_off = j.in_off;
      b0.c = j.c;
      b0.out_off = j.out_off;
      b0.cap = j.cap;
      if ((rc = ensure(g.ibufs, sizeof(ZesInfBuf) * 2))) return rc;
      hipLaunchKernelGGL(k_inf_set_table1, dim3(1), dim3(64), 0, g.stream, b0, b0, (ZesInfBuf*)g.ibufs.p, (uint32_t*)nullptr, 0u);
      Timed t("k_inf_decode_seq");
      hipLaunchKernelGGL(k_inf_decode, dim3(1), dim3(64), 0, g.stream, d_in, d_out, (const ZesInfBuf*)g.ibufs.p, (ZesRes*)g.res.p,
                         (uint64_t*)g.resume.p);
    }
    if ((rc = read_res(&hr))) return rc;
#ifdef WD_PROFILE
    {
      unsigned long long h[8];
      HIPCHK(hipMemcpy(h, g.dbg.p, 64, hipMemcpyDeviceToHost));
      const double n = (double)std::max<unsigned long long>(h[5], 1);
      fprintf(stderr, "zes wave decoder: %llu tokens, cycles per token: lit/len entry %.1f, literal path %.1f, distance entry %.1f, copy %.1f, flush test %.1f; kernel %.1f\n",
              h[5], h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[6] / n);
    }
#endif
    if (hr.status == 0) {
      j.tier = 3;
      j.out_len = hr.out_len;
      j.status = hr.out_len > j.cap ? ZES_E_NOSPACE : ZES_OK;
      return ZES_OK;
    }
    have_resume = true;
  }
  // T4: exact restatement from the failing block on
  {
    Timed t("k_inf_exact");
    hipLaunchKernelGGL(k_inf_exact, dim3(1), dim3(64), 0, g.stream, d_in, j.in_off, j.c, d_out, j.out_off, j.cap,
                       have_resume ? (const uint64_t*)g.resume.p : (const uint64_t*)nullptr, (ZesRes*)g.res.p);
  }
  if ((rc = read_res(&hr))) return rc;
  j.tier = 4;
  j.out_len = hr.out_len;
  j.status = hr.status;
  return ZES_OK;
}

// A reference-made stream too long for 32-bit bit positions (c >= 512 MiB), or any stream when ZES_F_PIECES asks for it
// (testing aid): T1 piece by piece.  A piece = the blocks that start inside the next `piece` bytes behind the end of
// the piece before; it is handed over with enough bytes behind it for its last block and the header behind that.
constexpr uint64_t T1_PIECE = 256ull << 20;
constexpr uint64_t T1_PIECE_SLACK = 1ull << 20;
int inflate_pieces(const uint8_t* d_in, uint8_t* d_out, InfJob& j, uint32_t flags) {
  uint64_t piece = T1_PIECE;
  if (flags & ZES_F_PIECES) piece = 1ull << 20;  // testing aid: 1 MiB pieces
  uint64_t pos_bit = 16, total = 0, blocks = 0;
  for (uint32_t round = 0; round < (1u << 20); round++) {
    uint64_t byte0 = (pos_bit >> 3) & ~15ull;
    if (byte0 >= 16) byte0 -= 16;  // (the block search starts 16 bits into a buffer: keep the piece's first block behind that)
    const uint64_t rel = pos_bit - 8 * byte0;
    const uint64_t pc = std::min<uint64_t>(j.c - byte0, piece + T1_PIECE_SLACK);
    const uint64_t done_bytes = blocks * ZES_BLK;
    RangeRes rr;
    int rc = inflate_t1_range(d_in, j.in_off + byte0, pc, rel, rel + piece * 8, true, d_out, j.out_off + std::min(done_bytes, j.cap),
                              j.cap > done_bytes ? j.cap - done_bytes : 0, flags, &rr);
    if (rc) return rc;
    if (!rr.handled || rr.nblocks == 0) return ZES_OK;  // not a clean chain: the other tiers decide
    blocks += rr.nblocks;
    total += rr.out_len;
    pos_bit = 8 * byte0 + rr.end_bit;
    if (rr.final_block) {
      j.tier = 1;
      j.out_len = total;
      j.status = total > j.cap ? ZES_E_NOSPACE : ZES_OK;
      return ZES_OK;
    }
    if (pos_bit >= j.c * 8) return ZES_OK;
  }
  return ZES_OK;
}

// All jobs of a call: T1 in groups, then the stragglers one by one.  jobs[i].status must be ZES_OK
// for the buffers to decode (anything else is left untouched).  firsts[i] = first byte of buffer i.
int inflate_jobs(const uint8_t* d_in, uint8_t* d_out, std::vector<InfJob>& jobs, const uint8_t* firsts, uint32_t flags) {
  g.last_tier = 0;
  g.sv_ok = false;  // (a survivor list serves the call that made it: the bytes behind a pointer may have changed since)
  std::vector<uint32_t> ids;
  std::vector<uint32_t> todo;
  std::vector<uint32_t> small;  // buffers T1 does not take and whose first byte is still on the device
  for (uint32_t i = 0; i < jobs.size(); i++) {
    InfJob& j = jobs[i];
    j.out_len = 0;
    j.tier = 0;
    if (j.status) continue;
    if (j.c == 0 || (firsts && (firsts[i] & 15u) != 8u)) {  // src/zlib.ts:13-16
      j.status = ZES_E_NOT_DEFLATE;
      j.tier = -1;
      continue;
    }
    todo.push_back(i);
    if (t1_eligible(j, flags)) ids.push_back(i);
    else if (!firsts) small.push_back(i);
  }
  int rc;
  // first bytes nobody supplied: T1 gets them back with its counters; the buffers T1 does not take are
  // gathered here (device kernel + one read-back per group)
  for (size_t g0 = 0; g0 < small.size(); g0 += INF_GROUP) {
    const uint32_t nb = (uint32_t)std::min<size_t>(INF_GROUP, small.size() - g0);
    if ((rc = ensure(g.ibufs, (size_t)INF_GROUP * 8 + INF_GROUP))) return rc;
    uint64_t* ho = (uint64_t*)((uint8_t*)g.pinned + PIN_UP);
    for (uint32_t k = 0; k < nb; k++) ho[k] = jobs[small[g0 + k]].in_off;
    uint8_t* dfirst = (uint8_t*)g.ibufs.p + (size_t)INF_GROUP * 8;
    HIPCHK(hipMemcpyAsync(g.ibufs.p, ho, (size_t)nb * 8, hipMemcpyHostToDevice, g.stream));
    hipLaunchKernelGGL(k_inf_first_bytes, dim3((nb + 255) / 256), dim3(256), 0, g.stream, d_in, (const uint64_t*)g.ibufs.p, dfirst, nb);
    HIPCHK(hipMemcpyAsync(g.pinned, dfirst, nb, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    for (uint32_t k = 0; k < nb; k++)
      if ((((const uint8_t*)g.pinned)[k] & 15u) != 8u) {
        jobs[small[g0 + k]].status = ZES_E_NOT_DEFLATE;
        jobs[small[g0 + k]].tier = -1;
      }
  }
  for (size_t g0 = 0; g0 < ids.size(); g0 += INF_GROUP) {
    const uint32_t nb = (uint32_t)std::min<size_t>(INF_GROUP, ids.size() - g0);
    if ((rc = inflate_t1_group(d_in, d_out, jobs.data(), ids.data() + g0, nb, firsts == nullptr, flags))) return rc;
  }
  // streams too long for the block-parallel tier's 32-bit bit positions (and ZES_F_PIECES): the same tier, piece by piece
  for (uint32_t i : todo) {
    InfJob& j = jobs[i];
    if (j.tier != 0 || j.status != ZES_OK || (flags & ZES_F_NO_FASTPATH)) continue;
    if (!(j.c >= (1ull << 29) || ((flags & ZES_F_PIECES) && j.c >= 64))) continue;
    if (firsts == nullptr) {  // the CM nibble (src/zlib.ts:13-16) was not checked on the way for this one
      uint8_t fb = 0;
      HIPCHK(hipMemcpyAsync(g.pinned, d_in + j.in_off, 1, hipMemcpyDeviceToHost, g.stream));
      HIPCHK(hipStreamSynchronize(g.stream));
      fb = *(const uint8_t*)g.pinned;
      if ((fb & 15u) != 8u) {
        j.status = ZES_E_NOT_DEFLATE;
        j.tier = -1;
        continue;
      }
    }
    if ((rc = inflate_pieces(d_in, d_out, j, flags))) return rc;
  }
  // Many streams the block-parallel tier left over (a batch of another encoder's streams): one serial wavefront
  // per stream, all at once — 512 of them run side by side, where the per-buffer tiers would take the streams one
  // after the other.  Streams that fail here go on to the per-buffer tiers.
  {
    std::vector<uint32_t> rest;
    for (uint32_t i : todo)
      if (jobs[i].tier == 0 && jobs[i].status == ZES_OK && jobs[i].c >= 3 && jobs[i].c < SERIAL_BATCH_MAX_C) rest.push_back(i);
    if (rest.size() >= SERIAL_BATCH_MIN_JOBS) {
      for (size_t g0 = 0; g0 < rest.size(); g0 += INF_GROUP) {
        const uint32_t nb = (uint32_t)std::min<size_t>(INF_GROUP, rest.size() - g0);
        ZesInfBuf* hb = (ZesInfBuf*)((uint8_t*)g.pinned + PIN_UP);
        memset(hb, 0, sizeof(ZesInfBuf) * nb);
        for (uint32_t k = 0; k < nb; k++) {
          const InfJob& j = jobs[rest[g0 + k]];
          hb[k].in_off = j.in_off;
          hb[k].c = j.c;
          hb[k].out_off = j.out_off;
          hb[k].cap = j.cap;
        }
        if ((rc = ensure(g.ibufs, sizeof(ZesInfBuf) * nb))) return rc;
        if ((rc = ensure(g.res, sizeof(ZesRes) * nb))) return rc;
        if ((rc = ensure(g.resume, (size_t)16 * nb))) return rc;
        HIPCHK(hipMemcpyAsync(g.ibufs.p, hb, sizeof(ZesInfBuf) * nb, hipMemcpyHostToDevice, g.stream));
        {
          Timed t("k_inf_decode_seq");
          hipLaunchKernelGGL(k_inf_decode, dim3(nb), dim3(64), 0, g.stream, d_in, d_out, (const ZesInfBuf*)g.ibufs.p, (ZesRes*)g.res.p,
                             (uint64_t*)g.resume.p);
        }
        ZesRes* hres = (ZesRes*)((uint8_t*)g.pinned + 128 * 1024);
        HIPCHK(hipMemcpyAsync(hres, g.res.p, sizeof(ZesRes) * nb, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
        for (uint32_t k = 0; k < nb; k++) {
          if (hres[k].status != 0) continue;
          InfJob& j = jobs[rest[g0 + k]];
          j.tier = 3;
          j.out_len = hres[k].out_len;
          j.status = hres[k].out_len > j.cap ? ZES_E_NOSPACE : ZES_OK;
        }
      }
    }
  }
  if (!(flags & ZES_F_NO_FASTPATH)) {
    {
      // the stored-blocks walk is a launch and a read-back per buffer: with several buffers left, only those whose first
      // block IS a stored one (BTYPE, bits 1-2 of the byte behind the zlib header; gathered by one launch per group) try it
      // (a buffer the block-parallel tier has looked at comes with its first BTYPE: no attempt, and no launch to find out)
      std::vector<uint32_t> st;
      for (uint32_t i : todo)
        if (jobs[i].tier == 0 && jobs[i].status == ZES_OK && jobs[i].c >= STORED_MIN_C && (jobs[i].btype0 < 0 || jobs[i].btype0 == 0)) st.push_back(i);
      std::vector<char> want(st.size(), 1);
      if (st.size() >= 2) {
        for (size_t g0 = 0; g0 < st.size(); g0 += INF_GROUP) {
          const uint32_t nb = (uint32_t)std::min<size_t>(INF_GROUP, st.size() - g0);
          if ((rc = ensure(g.ibufs, (size_t)INF_GROUP * 8 + INF_GROUP))) return rc;
          uint64_t* ho = (uint64_t*)((uint8_t*)g.pinned + PIN_UP);
          for (uint32_t k = 0; k < nb; k++) ho[k] = jobs[st[g0 + k]].in_off + 2u;
          uint8_t* dfirst = (uint8_t*)g.ibufs.p + (size_t)INF_GROUP * 8;
          HIPCHK(hipMemcpyAsync(g.ibufs.p, ho, (size_t)nb * 8, hipMemcpyHostToDevice, g.stream));
          hipLaunchKernelGGL(k_inf_first_bytes, dim3((nb + 255) / 256), dim3(256), 0, g.stream, d_in, (const uint64_t*)g.ibufs.p, dfirst, nb);
          HIPCHK(hipMemcpyAsync(g.pinned, dfirst, nb, hipMemcpyDeviceToHost, g.stream));
          HIPCHK(hipStreamSynchronize(g.stream));
          for (uint32_t k = 0; k < nb; k++) want[g0 + k] = ((((const uint8_t*)g.pinned)[k] >> 1) & 3u) == 0u;
        }
      }
      for (size_t q = 0; q < st.size(); q++)
        if (want[q] && (rc = inflate_stored(d_in, d_out, jobs[st[q]]))) return rc;
    }
    // (ZES_SEG_PIECE_MB set: every stream goes piece by piece, for tests)
    static const bool force_pieces = getenv("ZES_SEG_PIECE_MB") != nullptr;
    std::vector<uint32_t> segs;
    for (uint32_t i : todo)
      if (jobs[i].tier == 0 && jobs[i].status == ZES_OK && jobs[i].c >= SEG_MIN_C && jobs[i].c < SEG_PIECES_MIN_C && !force_pieces) segs.push_back(i);
    if (!segs.empty() && (rc = inflate_segments(d_in, d_out, jobs.data(), segs))) return rc;
    // Longer streams: the same tier, piece by piece.  Not only the ones beyond 32-bit bit positions (512 MiB): the tier
    // takes at most SEG_BUCKETS work items per buffer, so a long stream's candidates are thinned, every item is then
    // several blocks long, and all but the first of them are decoded by a lone wave — 155 MiB of zlib -1 text
    // (400 MiB): 88 ms in one go, 12.7 ms in five pieces.
    for (uint32_t i : todo)
      if (jobs[i].tier == 0 && jobs[i].status == ZES_OK && (jobs[i].c >= SEG_PIECES_MIN_C || (force_pieces && jobs[i].c >= SEG_MIN_C)) &&
          (rc = inflate_segments_pieces(d_in, d_out, jobs[i])))
        return rc;
  }
  int worst = 0;
  for (uint32_t i : todo) {
    if (jobs[i].tier == 0 && jobs[i].status == ZES_OK && (rc = inflate_slow(d_in, d_out, jobs[i]))) return rc;
    worst = std::max(worst, jobs[i].tier);
  }
  collect_times();
  g.last_tier = worst;
  return ZES_OK;
}

// One buffer at d_in+in_off (16-byte aligned), result at d_out+out_off (16-byte aligned).
// Returns the reference-equivalent status; *out_len = bytes produced (or needed on NOSPACE).
int inflate_one(const uint8_t* d_in, uint64_t in_off, uint64_t c, uint8_t* d_out, uint64_t out_off, uint64_t cap,
                uint64_t* out_len, uint32_t flags, int first_byte /* -1: still on the device */) {
  std::vector<InfJob> jobs(1);
  jobs[0] = InfJob{in_off, c, out_off, cap, 0, ZES_OK, 0};
  const uint8_t fb = (uint8_t)first_byte;
  int rc = inflate_jobs(d_in, d_out, jobs, first_byte < 0 ? nullptr : &fb, flags);
  if (rc) return rc;
  *out_len = jobs[0].out_len;
  return jobs[0].status;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------
extern "C" {

const char* zes_strerror(int status) {
  switch (status) {
    case ZES_OK: return "ok";
    case ZES_E_NOT_DEFLATE: return "Not compressed by deflate";
    case ZES_E_BTYPE3: return "Not supported BTYPE : 3";
    case ZES_E_CORRUPT: return "Data is corrupted";
    case ZES_E_INSUFFICIENT: return "Data length is insufficient";
    case ZES_E_LACK: return "Lack of data length";
    case ZES_E_NOSPACE: return "zes: output capacity too small";
    case ZES_E_DEVICE: return "zes: HIP device error (no gfx950 device or runtime failure)";
    case ZES_E_ARG: return "zes: bad argument";
    case ZES_E_NOTRANGE: return "zes: not a clean chain of reference-made blocks in this range";
    default: return "zes: unknown status";
  }
}

int zes_init(int device) {
  UseDev ud(0);
  std::lock_guard<std::mutex> lk(g_mu);
  return init_locked(device);
}

int zes_init_devices(int n) {
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return ZES_E_DEVICE;
  // ZES_OVERSUBSCRIBE: more contexts than devices, context i on device i % devices (exercises the multi-device paths on a one-GPU box)
  const bool over = getenv("ZES_OVERSUBSCRIBE") != nullptr;
  if (n <= 0) n = have;
  if (n > ZES_MAX_DEV || (n > have && !over)) return ZES_E_ARG;
  std::lock_guard<std::mutex> cfg(g_cfg_mu);
  for (int i = 0; i < n; i++) {
    std::lock_guard<std::mutex> lk(g_mus[i]);
    const int dev = i % have;
    if (g_ctx[i].ready ? g_ctx[i].device != dev : (g_ctx[i].want >= 0 && g_ctx[i].want != dev)) return ZES_E_ARG;  // bound elsewhere already
    g_ctx[i].want = dev;
  }
  if (n > g_nctx.load()) g_nctx.store(n);
  for (int i = 0; i < n; i++) {  // bring every context up now: a first batch should not pay for it
    UseDev ud(i);
    std::lock_guard<std::mutex> lk(g_mu);
    const int rc = init_locked(-1);
    if (rc) return rc;
  }
  return ZES_OK;
}

static int shutdown_one(void);
int zes_shutdown(void) {
  std::lock_guard<std::mutex> cfg(g_cfg_mu);
  int rc = ZES_OK;
  for (int i = 0; i < ZES_MAX_DEV; i++) {
    t_dev = i;  // (not a routed call: every context in turn)
    const int r = shutdown_one();
    if (r && !rc) rc = r;
    g_ctx[i].want = -1;
  }
  t_dev = 0;
  g_nctx.store(1);
  return rc;
}

// every pooled device buffer of the current context
static void free_scratch_locked() {
  DevBuf* all[] = {&g.bufs, &g.blks, &g.idx_a, &g.idx_b, &g.sdelta, &g.tmask, &g.mlist, &g.hists, &g.codes, &g.hdrs, &g.adler, &g.res, &g.order, &g.surv, &g.vlong, &g.segfail, &g.symoff, &g.cand,
                   &g.cand_sorted, &g.counters, &g.cres, &g.map, &g.resume, &g.dbg, &g.ibufs, &g.ibufs2, &g.mvlist, &g.scratch, &g.st_in, &g.st_out,
                   &g.sres, &g.maps, &g.seglist, &g.segprefix, &g.wins, &g.sym16, &g.segorder, &g.segjobs, &g.pw16, &g.gwins, &g.seglive, &g.segouts};
  for (DevBuf* b : all) {
    if (b->p) (void)hipFree(b->p);
    b->p = nullptr;
    b->cap = 0;
  }
}

int zes_trim(void) {
  std::lock_guard<std::mutex> cfg(g_cfg_mu);
  const int keep = t_dev;
  for (int i = 0; i < ZES_MAX_DEV; i++) {
    t_dev = i;  // (not a routed call: every context in turn)
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g.ready) continue;
    (void)hipSetDevice(g.device);
    (void)hipStreamSynchronize(g.stream);
    free_scratch_locked();  // (the scan's constant table, g.kraft, stays)
  }
  t_dev = keep;
  return ZES_OK;
}

static int shutdown_one(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g.ready) return ZES_OK;
  (void)hipSetDevice(g.device);
  (void)hipStreamSynchronize(g.stream);
  free_scratch_locked();
  if (g.kraft.p) (void)hipFree(g.kraft.p);
  g.kraft.p = nullptr;
  g.kraft.cap = 0;
  if (g.pinned) (void)hipHostFree(g.pinned);
  g.pinned = nullptr;
  if (g.mirror) (void)hipHostFree(g.mirror);
  g.mirror = nullptr;
  g_side_up.shutdown();
  g_side_down.shutdown();
  g_up.release();
  g_down.release();
  for (hipEvent_t e : g.event_pool) (void)hipEventDestroy(e);
  g.event_pool.clear();
  (void)hipStreamDestroy(g.stream);
  (void)hipStreamDestroy(g.cs_in);
  (void)hipStreamDestroy(g.cs_out);
  for (int k = 0; k < 2; k++) (void)hipEventDestroy(g.ev_up[k]);
  for (int k = 0; k < 2; k++) (void)hipEventDestroy(g.ev_rng[k]);
  (void)hipEventDestroy(g.ev_k);
  (void)hipStreamDestroy(g.s_adler);
  (void)hipEventDestroy(g.ev_a0);
  (void)hipEventDestroy(g.ev_a1);
  g.s_adler = nullptr;
  g.stream = g.cs_in = g.cs_out = nullptr;
  g.ready = false;
  return ZES_OK;
}

int zes_host_alloc(uint64_t n, void** p) {
  UseDev ud(0);  // (page-locked memory belongs to the process: always through context 0)
  if (!p) return ZES_E_ARG;
  *p = nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  HIPCHK(hipHostMalloc(p, n ? n : 1, hipHostMallocPortable));  // (page-locked for every device the library drives)
  return ZES_OK;
}

int zes_host_free(void* p) {
  if (!p) return ZES_OK;
  // page-locked memory belongs to the process, not to a stream: it can be given back after zes_shutdown too (a JS
  // finalizer may run that late), and hipHostFree waits by itself for device work that still uses the block
  std::lock_guard<std::mutex> lk(g_mu);
  if (g.ready) (void)hipSetDevice(g.device);
  HIPCHK(hipHostFree(p));
  return ZES_OK;
}

int zes_device_info(char* name, int cap, int* cus, uint64_t* hbm_bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  if (name && cap > 0) snprintf(name, (size_t)cap, "%s", g.arch);
  if (cus) *cus = g.cus;
  if (hbm_bytes) *hbm_bytes = g.hbm;
  return ZES_OK;
}

int zes_deflate_bound(uint64_t n, uint64_t* cap) {
  if (!cap) return ZES_E_ARG;
  *cap = deflate_bound(n);
  return ZES_OK;
}

int zes_deflate_batch_dev(const uint8_t* d_in, const uint64_t* in_off, const uint64_t* in_len, uint8_t* d_out,
                          const uint64_t* out_off, const uint64_t* out_cap, uint64_t* out_len, int32_t* status, uint32_t count) {
  ROUTE_DEV(d_in, d_out);
  if (!in_off || !in_len || !out_off || !out_cap || !out_len || !status) return ZES_E_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  return deflate_batch_core(d_in, in_off, in_len, d_out, out_off, out_cap, out_len, status, count);
}

int zes_deflate_dev(const uint8_t* d_in, uint64_t n, uint8_t* d_out, uint64_t cap, uint64_t* out_len) {
  ROUTE_DEV(d_in, d_out);
  if (!out_len) return ZES_E_ARG;
  uint64_t zero = 0;
  int32_t st = 0;
  int rc = zes_deflate_batch_dev(d_in, &zero, &n, d_out, &zero, &cap, out_len, &st, 1);
  return rc ? rc : st;
}

// Host deflate of a large buffer in pieces of 256 blocks (one workgroup per CU), so that the three legs overlap: while
// the kernels work on piece k, piece k+1 crosses PCIe on one copy stream and the finished bytes of piece k-1 go back
// on another.  A piece is a block range (the machinery of zes_deflate_range_dev): it is emitted straight at its place
// in the output stream — bit offset = all bits before it, known when the piece before has finished — and continues
// the last dword of the piece before (ZES_BUF_CONT).  78 9C and the Adler-32 of the whole input (combined from the
// pieces': src/adler32.ts:1-10 is associative) are put into the caller's buffer by the host.  Same bytes as the
// one-pass path (tests/test_gpu_configs.py::test_pipelined_host_calls).
struct SettleGuard {  // no side task may outlive the frame whose variables it uses
  std::future<int>&a, &b;
  ~SettleGuard() {
    if (a.valid()) a.wait();
    if (b.valid()) b.wait();
  }
};
constexpr uint64_t PIPE_PIECE = 256ull * ZES_BLK;  // 32 MiB
constexpr uint64_t PIPE_MIN = PIPE_PIECE + PIPE_PIECE / 2;
constexpr uint64_t PIPE_HALO = 4096;  // a piece's match finder reads up to 258 bytes behind it: uploads are cut this far behind the piece ends
static int deflate_host_pipelined(const uint8_t* in, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* out_len) {
  int rc;
  const uint64_t bound = deflate_bound(n);
  if ((rc = ensure(g.st_in, n + 64))) return rc;
  if ((rc = ensure(g.st_out, bound + 64))) return rc;
  const uint8_t* d_in = (const uint8_t*)g.st_in.p;
  uint8_t* d_out = (uint8_t*)g.st_out.p;
  const uint32_t np = (uint32_t)((n + PIPE_PIECE - 1) / PIPE_PIECE);
  auto cut = [&](uint32_t k) { return k == 0 ? 0ull : k >= np ? n : std::min<uint64_t>(n, (uint64_t)k * PIPE_PIECE + PIPE_HALO); };
  auto up = [&](uint32_t k) -> int {  // upload k: bytes [cut(k), cut(k+1))
    int r = upload((uint8_t*)g.st_in.p + cut(k), in + cut(k), cut(k + 1) - cut(k), g.cs_in);
    if (r) return r;
    HIPCHK(hipEventRecord(g.ev_up[k & 1], g.cs_in));
    return ZES_OK;
  };
  const bool pdbg = getenv("ZES_PIPE_DBG") != nullptr;
  const auto t00 = std::chrono::steady_clock::now();
  auto stamp = [&](const char* what, uint32_t k) {
    if (pdbg) fprintf(stderr, "  pipe %-14s %u  %.3f ms\n", what, k, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t00).count());
  };
  if ((rc = up(0))) return rc;
  stamp("up done", 0);
  uint64_t pos = 16;       // bit position of the next piece in the output stream (behind 78 9C)
  uint64_t sent = 0;       // bytes of the output already on their way to the caller
  uint64_t s1 = 1, s2 = 0;  // Adler-32 of the pieces so far
  const bool out_fits = cap >= bound;  // (else: count first, the caller gets the size needed)
  uint64_t pend_lo = 0, pend_hi = 0;   // bytes of the output finished by the piece before, not yet sent
  std::future<int> f_up, f_down;
  SettleGuard guard{f_up, f_down};
  auto settle = [&](std::future<int>& f) { return f.valid() ? f.get() : (int)ZES_OK; };
  for (uint32_t k = 0; k < np; k++) {
    const uint64_t lo = (uint64_t)k * PIPE_PIECE, len = std::min<uint64_t>(PIPE_PIECE, n - lo);
    const uint64_t readable = std::min<uint64_t>(n - lo, len + 258);
    const uint64_t o_off = (pos >> 7) << 4;  // 16-byte aligned byte offset; the piece starts start_bit bits into it
    const uint32_t sbit = (uint32_t)(pos - o_off * 8);
    const uint64_t o_cap = bound + 64 - o_off;
    const uint32_t fl = ZES_BUF_RANGE | (k + 1 < np ? ZES_BUF_NOTFINAL : 0u) | (k ? ZES_BUF_CONT : 0u);
    uint64_t bits = 0, in_off = lo;
    int32_t st = 0;
    uint32_t ad = 1;
    if ((rc = settle(f_up))) {  // (piece k is up, or on its way with its event recorded)
      (void)settle(f_down);
      return rc;
    }
    HIPCHK(hipStreamWaitEvent(g.stream, g.ev_up[k & 1], 0));
    rc = deflate_batch_core(d_in, &in_off, &len, d_out, &o_off, &o_cap, &bits, &st, 1, &readable, &fl, &ad, &sbit, true);
    if (rc || st) {
      (void)settle(f_down);
      return rc ? rc : st;
    }
    // beside the kernels: the next piece up, the bytes the piece before finished down
    stamp("launched", k);
    if (k + 1 < np) f_up = g_side_up.submit([&up, &stamp, k] { int r = up(k + 1); stamp("up done", k + 1); return r; });
    if (out_fits && pend_hi > pend_lo) {
      if ((rc = settle(f_down))) return rc;
      const uint64_t a = pend_lo, b = pend_hi;
      f_down = g_side_down.submit([=, &stamp] { int r = download(out + a, d_out + a, b - a, g.cs_out, false); stamp("down issued", k); return r; });
      sent = pend_hi;
    }
    if (hipStreamSynchronize(g.stream) != hipSuccess) {
      (void)settle(f_up);
      (void)settle(f_down);
      return ZES_E_DEVICE;
    }
    collect_times();
    stamp("kernels done", k);
    const ZesRes* r = (const ZesRes*)g.pinned;
    if (r[0].status) {
      (void)settle(f_up);
      (void)settle(f_down);
      return r[0].status;
    }
    bits = r[0].out_len;
    ad = r[0].aux;
    pos += bits;
    const uint64_t a1 = ad & 0xFFFFu, a2 = ad >> 16;
    s2 = (s2 + a2 + (len % 65521u) * ((s1 + 65520u) % 65521u)) % 65521u;
    s1 = (s1 + a1 + 65520u) % 65521u;
    pend_lo = sent;
    pend_hi = (k + 1 < np) ? (pos >> 3) : ((pos + 7) >> 3);  // whole bytes; the last piece's padded end
  }
  if ((rc = settle(f_down))) return rc;
  const uint64_t raw_end = (pos + 7) >> 3, total = raw_end + 4;
  *out_len = total;
  if (total > cap) return ZES_E_NOSPACE;
  if (!out_fits) {  // a capacity below the bound that still holds the result: one copy now
    if ((rc = download(out, d_out, raw_end, g.cs_out, true))) return rc;
  } else {
    if (pend_hi > pend_lo && (rc = download(out + pend_lo, d_out + pend_lo, pend_hi - pend_lo, g.cs_out, false))) return rc;
    HIPCHK(hipStreamSynchronize(g.cs_out));
  }
  stamp("all down", np);
  out[0] = 0x78;  // src/zlib.ts:29-34
  out[1] = 0x9C;
  const uint32_t adl = (uint32_t)((s2 << 16) | s1);
  for (int k = 0; k < 4; k++) out[raw_end + k] = (uint8_t)(adl >> (24 - 8 * k));  // big-endian trailer (src/zlib.ts:37-40)
  return ZES_OK;
}

int zes_deflate(const uint8_t* in, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* out_len) {
  UseDev ud(route_host());
  if (!out_len || (!in && n) || !out) return ZES_E_ARG;
  *out_len = 0;
  if (deflate_throws(n)) return ZES_E_CORRUPT;
  const uint64_t bound = deflate_bound(n);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = init_locked(-1);
    if (rc) return rc;
    if (n >= PIPE_MIN && !getenv("ZES_NO_PIPELINE")) return deflate_host_pipelined(in, n, out, cap, out_len);
    if ((rc = ensure(g.st_in, n + 64))) return rc;
    if ((rc = ensure(g.st_out, bound + 64))) return rc;
    if ((rc = upload((uint8_t*)g.st_in.p, in, n))) return rc;
    uint64_t zero = 0, dl = 0;
    int32_t st = 0;
    rc = deflate_batch_core((const uint8_t*)g.st_in.p, &zero, &n, (uint8_t*)g.st_out.p, &zero, &bound, &dl, &st, 1);
    if (rc) return rc;
    if (st) return st;
    *out_len = dl;
    if (dl > cap) return ZES_E_NOSPACE;
    return download(out, (const uint8_t*)g.st_out.p, dl);
  }
}

int zes_inflate_dev(const uint8_t* d_in, uint64_t c, uint8_t* d_out, uint64_t cap, uint64_t* out_len, uint32_t flags) {
  host_lap("(outside the library)");
  ROUTE_DEV(d_in, d_out);
  if (!out_len) return ZES_E_ARG;
  if ((((uintptr_t)d_in) & 15u) || (((uintptr_t)d_out) & 15u)) return ZES_E_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  return inflate_one(d_in, 0, c, d_out, 0, cap, out_len, flags, -1);
}

int zes_inflate_batch_dev(const uint8_t* d_in, const uint64_t* in_off, const uint64_t* in_len, uint8_t* d_out,
                          const uint64_t* out_off, const uint64_t* out_cap, uint64_t* out_len, int32_t* status, uint32_t count,
                          uint32_t flags) {
  ROUTE_DEV(d_in, d_out);
  if (!in_off || !in_len || !out_off || !out_cap || !out_len || !status) return ZES_E_ARG;
  if ((((uintptr_t)d_in) & 15u) || (((uintptr_t)d_out) & 15u)) return ZES_E_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  for (uint32_t i = 0; i < count; i++) status[i] = ((in_off[i] & 15u) || (out_off[i] & 15u)) ? ZES_E_ARG : ZES_OK;
  std::vector<InfJob> jobs(count);
  for (uint32_t i = 0; i < count; i++) jobs[i] = InfJob{in_off[i], in_len[i], out_off[i], out_cap[i], 0, status[i], 0};
  rc = inflate_jobs(d_in, d_out, jobs, nullptr, flags);  // the first bytes (CM nibble, src/zlib.ts:13) are read on the way
  if (rc) return rc;
  for (uint32_t i = 0; i < count; i++) {
    status[i] = jobs[i].status;
    out_len[i] = jobs[i].out_len;
  }
  return ZES_OK;
}

// Host inflate of a long reference-made stream in pieces, so that the three legs overlap: the stream is cut into equal
// bit ranges (as shard.inflate_split cuts it over GPUs); while the block-parallel tier decodes the blocks that start in
// range k (inflate_t1_range), a helper thread sends range k+1 up on one copy stream and the output of range k-1 down on
// another.  The ranges must chain (each one's first block where the one before ended, full blocks everywhere but at
// the end, BFINAL last); anything else — another encoder's stream, a false block start — and the call starts over on
// the one-pass path below, which has every tier.  *done = false then.
constexpr uint64_t PIPE_IN_MIN = 8ull << 20;
static int inflate_host_pipelined(const uint8_t* in, uint64_t c, uint8_t* out, uint64_t cap, uint64_t* out_len, uint32_t flags,
                                  zes_alloc_fn alloc, void* user, bool* done) {
  *done = false;
  int rc;
  // Piece boundaries (multiples of 64 KiB).  A piece costs ~0.3 ms of launches and read-backs whatever its size, so
  // pieces of up to 32 MiB keep pace with their uploads; what cannot overlap anything is the LAST piece's decode and
  // download: from 48 MiB on the last piece is a small one (64 MiB of incompressible data: 8 + 24 + 24 + 8).
  std::vector<uint64_t> B;
  {
    uint64_t pb = std::min<uint64_t>(32ull << 20, std::max<uint64_t>(4ull << 20, ((c / 2 + 65535) >> 16) << 16));
    if (const char* e = getenv("ZES_PIPE_PIECE_MB")) pb = std::max<uint64_t>(1, strtoull(e, nullptr, 10)) << 20;  // (measurements)
    uint64_t last = 0;
    if (c >= (48ull << 20)) last = std::min<uint64_t>(8ull << 20, ((c / 8) >> 16) << 16);
    // ... and the FIRST piece's upload and decode, before the first byte can come down: a small one as well, when the
    // output is long enough to pay for one piece more (measured, pinned memory, ms per call without / with it: 64 MiB of
    // text 2.25 / 1.96, 128 MiB 4.02 / 3.45, 64 MiB of random bytes 2.36 / 2.28 — but 48 MiB of text 1.59 / 1.80, of
    // random bytes 2.01 / 2.16: a piece costs ~0.3 ms of launches and read-backs)
    const uint64_t est_out = (cap && !alloc) ? cap : c * 4;
    uint64_t first = (est_out >= (56ull << 20) && c >= (12ull << 20)) ? std::min<uint64_t>(8ull << 20, ((c / 3) >> 16) << 16) : 0;
    if (const char* e = getenv("ZES_PIPE_FIRST_MB")) first = strtoull(e, nullptr, 10) << 20;  // (measurements)
    if (first + last + (4ull << 20) > c) first = 0;
    const uint64_t rest = c - last - first;
    const uint64_t nr = std::max<uint64_t>(1, (rest + pb - 1) / pb);
    const uint64_t each = (((rest + nr - 1) / nr + 65535) >> 16) << 16;
    B.push_back(0);
    if (first) B.push_back(first);
    for (uint64_t k = 1; k < nr; k++)
      if (k * each < rest) B.push_back(first + k * each);
    if (last && first + rest > B.back()) B.push_back(first + rest);
    B.push_back(c);
  }
  const uint32_t np = (uint32_t)B.size() - 1;
  if (np < 2) return ZES_OK;
  const uint64_t dcap = std::max<uint64_t>(alloc ? 0 : cap, std::max<uint64_t>(c * 4, 1 << 20));
  if ((rc = ensure(g.st_in, c + 64))) return rc;
  if ((rc = ensure(g.st_out, dcap + 64))) return rc;
  const uint8_t* d_in = (const uint8_t*)g.st_in.p;
  uint8_t* d_out = (uint8_t*)g.st_out.p;
  auto cut = [&](uint32_t k) { return k == 0 ? 0ull : k >= np ? c : std::min<uint64_t>(c, B[k] + T1_PIECE_SLACK); };
  auto up = [&](uint32_t k) -> int {
    int r = upload((uint8_t*)g.st_in.p + cut(k), in + cut(k), cut(k + 1) - cut(k), g.cs_in);
    if (r) return r;
    HIPCHK(hipEventRecord(g.ev_up[k & 1], g.cs_in));
    return ZES_OK;
  };
  {
    uint64_t cmax = 0;
    for (uint32_t k = 0; k < np; k++) cmax = std::max<uint64_t>(cmax, B[k + 1] - B[k] + T1_PIECE_SLACK + 64);
    if ((rc = range_reserve(cmax))) return rc;
  }
  if ((rc = up(0))) return rc;
  uint64_t blocks = 0, total = 0, prev_end = 16, pend_lo = 0, pend_hi = 0;
  bool final_seen = false, chain = true;
  std::future<int> f_up, f_down;
  SettleGuard guard{f_up, f_down};
  auto settle = [&](std::future<int>& f) { return f.valid() ? f.get() : (int)ZES_OK; };
  // compressible data or not (which form of the block decoder): by the whole call, not by a piece
  const bool two = c * 10 < std::min<uint64_t>(dcap, cap ? cap : dcap) * 7;
  bool early = alloc && (flags & ZES_F_ALLOC_BOUND);  // the allocator takes an upper estimate (include/zes.h)
  RangePend pend[2];
  uint64_t byte0s[2] = {0, 0};
  auto begin = [&](uint32_t k) -> int {  // (range k is up, or on its way with its event recorded)
    HIPCHK(hipStreamWaitEvent(g.stream, g.ev_up[k & 1], 0));
    const uint64_t lo_bit = k == 0 ? 16 : B[k] * 8, own_bit = B[k + 1] * 8;
    uint64_t byte0 = (lo_bit >> 3) & ~15ull;
    if (byte0 >= 16) byte0 -= 16;
    byte0s[k & 1] = byte0;
    const uint64_t pc = std::min<uint64_t>(c - byte0, (own_bit >> 3) - byte0 + T1_PIECE_SLACK);
    return range_begin((int)(k & 1), pend[k & 1], d_in, byte0, pc, lo_bit - 8 * byte0, own_bit - 8 * byte0, k == 0, d_out, dcap, two, flags);
  };
  if (np > 1) f_up = g_side_up.submit([&up] { return up(1); });
  if ((rc = begin(0))) return rc;
  for (uint32_t k = 0; k < np && chain && !final_seen; k++) {
    if (k + 1 < np) {  // the next piece is enqueued before this one's results are waited for
      if ((rc = settle(f_up))) return rc;
      if (k + 2 < np) f_up = g_side_up.submit([&up, k] { return up(k + 2); });
      if ((rc = begin(k + 1))) return rc;
    }
    RangeRes rr;
    if ((rc = range_finish((int)(k & 1), pend[k & 1], &rr))) return rc;
    const uint64_t byte0 = byte0s[k & 1];
    if (!rr.handled) {
      chain = false;
    } else if (rr.nblocks) {
      if (rr.first_bit + 8 * byte0 != prev_end || total != blocks * ZES_BLK) {
        chain = false;
      } else {
        prev_end = rr.end_bit + 8 * byte0;
        pend_lo = std::min(total, cap);
        total += rr.out_len;
        pend_hi = std::min(total, cap);
        blocks += rr.nblocks;
        final_seen = rr.final_block;
        if (early && !out) {
          // the first piece is decoded: from its blocks' ratio an upper estimate of the whole result (5 % and two blocks on
          // top), asked for now, so that every piece's bytes can go down while the ones behind it are decoded
          const uint64_t in0 = std::max<uint64_t>(1, (prev_end + 7) / 8);
          const long double ratio = (long double)total / (long double)in0;
          uint64_t est = (uint64_t)((long double)c * ratio * 1.05L) + 2 * ZES_BLK;
          est = std::min<uint64_t>(std::max<uint64_t>(est, total), dcap);
          out = alloc(user, ZES_ALLOC_EARLY, est);
          if (out) {
            cap = est;
            pend_hi = std::min(total, cap);
          } else {
            early = false;  // "not now": the exact size, once, when it is known
          }
        }
        if (out && (!alloc || early) && pend_hi > pend_lo && !(final_seen || k + 1 == np)) {  // (the last piece's bytes go down below)
          if ((rc = settle(f_down))) return rc;
          const uint64_t a = pend_lo, b = pend_hi;
          f_down = g_side_down.submit([=] { return download(out + a, d_out + a, b - a, g.cs_out, false); });
          pend_lo = pend_hi = 0;
        }
      }
    }
  }
  const bool dbgp = getenv("ZES_DEBUG_PIPE") != nullptr;
  auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tp0 = dbgp ? tnow() : 0;
  if ((rc = settle(f_up))) return rc;
  if ((rc = settle(f_down))) return rc;
  HIPCHK(hipStreamSynchronize(g.cs_in));
  HIPCHK(hipStreamSynchronize(g.stream));  // (a piece enqueued ahead of a result that ended the loop)
  if (dbgp) fprintf(stderr, "zes pipe: loop done, settle+sync %.3f ms, np %u total %llu\n", tnow() - tp0, np, (unsigned long long)total);
  if (!chain || !final_seen || total > dcap) {
    HIPCHK(hipStreamSynchronize(g.cs_out));
    return ZES_OK;  // not this way: the one-pass path decides
  }
  *done = true;
  g.last_tier = 1;
  *out_len = total;
  if (alloc && early && out && total <= cap) {  // the estimate held: what is left to go down is the last piece
    if (pend_hi > pend_lo && (rc = download(out + pend_lo, d_out + pend_lo, pend_hi - pend_lo, g.cs_out, false))) return rc;
    HIPCHK(hipStreamSynchronize(g.cs_out));
    return ZES_OK;
  }
  if (alloc) {  // the caller allocates the exact result now that its size is known (or: the early estimate fell short)
    HIPCHK(hipStreamSynchronize(g.cs_out));
    const double ta = dbgp ? tnow() : 0;
    out = alloc(user, 0, total);
    if (!out) return ZES_E_ARG;
    const double tb = dbgp ? tnow() : 0;
    rc = download(out, d_out, total, g.cs_out, true);
    if (dbgp) fprintf(stderr, "zes pipe: alloc %.3f ms, download %.3f ms\n", tb - ta, tnow() - tb);
    return rc;
  }
  if (total > cap) {
    HIPCHK(hipStreamSynchronize(g.cs_out));
    return ZES_E_NOSPACE;
  }
  if (pend_hi > pend_lo && (rc = download(out + pend_lo, d_out + pend_lo, pend_hi - pend_lo, g.cs_out, false))) return rc;
  HIPCHK(hipStreamSynchronize(g.cs_out));
  return ZES_OK;
}

static int inflate_host(const uint8_t* in, uint64_t c, uint8_t* out, uint64_t cap, uint64_t* out_len, uint32_t flags,
                        bool size_only, zes_alloc_fn alloc, void* user) {
  if (!out_len || (!in && c)) return ZES_E_ARG;
  *out_len = 0;
  if (c == 0 || (in[0] & 15u) != 8u) return ZES_E_NOT_DEFLATE;  // src/zlib.ts:13-16, decided before the device is touched
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  if (c >= PIPE_IN_MIN && c < (1ull << 29) && !size_only && (out || alloc) && !(flags & (ZES_F_NO_FASTPATH | ZES_F_PIECES)) &&
      !getenv("ZES_NO_PIPELINE")) {
    bool done = false;
    rc = inflate_host_pipelined(in, c, out, cap, out_len, flags, alloc, user, &done);
    if (rc || done) return rc;
    *out_len = 0;
  }
  if ((rc = ensure(g.st_in, c + 64))) return rc;
  if ((rc = upload((uint8_t*)g.st_in.p, in, c))) return rc;
  // decode into pooled device memory: grow-and-retry like the reference's Uint8WriteStream
  uint64_t dcap = std::max<uint64_t>((size_only || alloc) ? 0 : cap, std::max<uint64_t>(c * 4, 1 << 20));
  for (int attempt = 0; attempt < 8; attempt++) {
    if ((rc = ensure(g.st_out, dcap + 64))) return rc;
    uint64_t n = 0;
    rc = inflate_one((const uint8_t*)g.st_in.p, 0, c, (uint8_t*)g.st_out.p, 0, dcap, &n, flags, c ? in[0] : 0);
    if (rc == ZES_E_NOSPACE && n > dcap) {
      dcap = n;
      continue;
    }
    if (rc) return rc;
    *out_len = n;
    if (size_only) return ZES_OK;
    if (alloc) {  // the caller allocates the exact result now that its size is known; still under the lock
      out = alloc(user, 0, n);
      if (!out) return ZES_E_ARG;
      cap = n;
    }
    if (n > cap) return ZES_E_NOSPACE;
    return download(out, (const uint8_t*)g.st_out.p, n);
  }
  return ZES_E_DEVICE;
}

int zes_inflate(const uint8_t* in, uint64_t c, uint8_t* out, uint64_t cap, uint64_t* out_len, uint32_t flags) {
  UseDev ud(route_host());
  return inflate_host(in, c, out, cap, out_len, flags, false, nullptr, nullptr);
}
int zes_inflate_size(const uint8_t* in, uint64_t c, uint64_t* n, uint32_t flags) {
  UseDev ud(route_host());
  return inflate_host(in, c, nullptr, 0, n, flags, true, nullptr, nullptr);
}
int zes_inflate_alloc(const uint8_t* in, uint64_t c, zes_alloc_fn alloc, void* user, uint64_t* out_len, uint32_t flags) {
  UseDev ud(route_host());
  if (!alloc) return ZES_E_ARG;
  return inflate_host(in, c, nullptr, 0, out_len, flags, false, alloc, user);
}

// ---- batch over host pointers: one arena up, the device batch, results down ----
static int deflate_batch_one(const uint8_t* const* in, const uint64_t* in_len, uint8_t* const* out, const uint64_t* out_cap,
                             uint64_t* out_len, int32_t* status, uint32_t count) {
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  std::vector<uint64_t> in_off(count), o_off(count), o_cap(count), dl(count);
  uint64_t tin = 0, tout = 0;
  for (uint32_t i = 0; i < count; i++) {
    if (!in[i] && in_len[i]) return ZES_E_ARG;
    in_off[i] = tin;
    tin += (in_len[i] + 64 + 15) & ~15ull;
    o_off[i] = tout;
    o_cap[i] = deflate_bound(in_len[i]);
    tout += (o_cap[i] + 15) & ~15ull;
  }
  if ((rc = ensure(g.st_in, tin + 64))) return rc;
  if ((rc = ensure(g.st_out, tout + 64))) return rc;
  for (uint32_t i = 0; i < count; i++)
    if (!deflate_throws(in_len[i]) && (rc = upload((uint8_t*)g.st_in.p + in_off[i], in[i], in_len[i]))) return rc;
  rc = deflate_batch_core((const uint8_t*)g.st_in.p, in_off.data(), in_len, (uint8_t*)g.st_out.p, o_off.data(), o_cap.data(), dl.data(), status, count);
  if (rc) return rc;
  for (uint32_t i = 0; i < count; i++) {
    out_len[i] = dl[i];
    if (status[i]) continue;
    if (dl[i] > out_cap[i] || !out[i]) {
      status[i] = out[i] ? ZES_E_NOSPACE : ZES_E_ARG;
      continue;
    }
    if ((rc = download(out[i], (const uint8_t*)g.st_out.p + o_off[i], dl[i]))) return rc;
  }
  return ZES_OK;
}

static int inflate_batch_alloc_one(const uint8_t* const* in, const uint64_t* in_len, zes_alloc_fn alloc, void* user, uint64_t* out_len,
                                   int32_t* status, uint32_t count, uint32_t flags) {
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  std::vector<uint64_t> in_off(count), o_off(count), o_cap(count);
  std::vector<uint8_t> firsts(count);
  uint64_t tin = 0;
  for (uint32_t i = 0; i < count; i++) {
    if (!in[i] && in_len[i]) return ZES_E_ARG;
    in_off[i] = tin;
    tin += (in_len[i] + 64 + 15) & ~15ull;
    firsts[i] = in_len[i] ? in[i][0] : 0;
    // first guess at the result size: a reference-made stream of c bytes rarely inflates beyond 4c; the retry below has exact sizes
    o_cap[i] = std::max<uint64_t>(in_len[i] * 4, 1 << 16);
    out_len[i] = 0;
    status[i] = ZES_OK;
  }
  if ((rc = ensure(g.st_in, tin + 64))) return rc;
  for (uint32_t i = 0; i < count; i++)
    if ((rc = upload((uint8_t*)g.st_in.p + in_off[i], in[i], in_len[i]))) return rc;
  std::vector<uint32_t> todo(count);
  for (uint32_t i = 0; i < count; i++) todo[i] = i;
  for (int attempt = 0; attempt < 8 && !todo.empty(); attempt++) {
    uint64_t tout = 0;
    std::vector<InfJob> jobs(todo.size());
    std::vector<uint8_t> fb(todo.size());
    for (size_t k = 0; k < todo.size(); k++) {
      const uint32_t i = todo[k];
      o_off[i] = tout;
      tout += (o_cap[i] + 15) & ~15ull;
      jobs[k] = InfJob{in_off[i], in_len[i], o_off[i], o_cap[i], 0, ZES_OK, 0};
      fb[k] = firsts[i];
    }
    if ((rc = ensure(g.st_out, tout + 64))) return rc;
    if ((rc = inflate_jobs((const uint8_t*)g.st_in.p, (uint8_t*)g.st_out.p, jobs, fb.data(), flags))) return rc;
    std::vector<uint32_t> again;
    for (size_t k = 0; k < todo.size(); k++) {
      const uint32_t i = todo[k];
      out_len[i] = jobs[k].out_len;
      status[i] = jobs[k].status;
      if (jobs[k].status == ZES_E_NOSPACE && jobs[k].out_len > o_cap[i]) {
        o_cap[i] = jobs[k].out_len;
        again.push_back(i);
        continue;
      }
      if (jobs[k].status) continue;
      uint8_t* dst = alloc(user, i, jobs[k].out_len);
      if (!dst) {
        status[i] = ZES_E_ARG;
        continue;
      }
      if ((rc = download(dst, (const uint8_t*)g.st_out.p + o_off[i], jobs[k].out_len))) return rc;
    }
    todo.swap(again);
  }
  for (uint32_t i : todo) status[i] = ZES_E_DEVICE;
  return ZES_OK;
}

// ---- a host batch over every device the library drives ----
// Size-balanced owner lists: the longest buffers first, each onto the lightest part so far (ties: the lower index) —
// the rule of shard.partition, so that a Node batch and a torch.distributed job cut the same work the same way.
int zes_partition(const uint64_t* sizes, uint32_t count, uint32_t parts, uint32_t* owner) {
  if ((!sizes || !owner) && count) return ZES_E_ARG;
  if (parts == 0) return ZES_E_ARG;
  std::vector<uint32_t> order(count);
  for (uint32_t i = 0; i < count; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return sizes[a] > sizes[b]; });
  std::vector<uint64_t> load(parts, 0);
  for (uint32_t i : order) {
    uint32_t best = 0;
    for (uint32_t r = 1; r < parts; r++)
      if (load[r] < load[best]) best = r;
    owner[i] = best;
    load[best] += sizes[i];
  }
  return ZES_OK;
}

int zes_device_count(void) {
  return g_nctx.load();
}

}  // extern "C"
// runs fn(ids) for every non-empty share, each on its own thread bound to its context; the first non-zero return wins
template <class F>
static int over_devices(const uint64_t* sizes, uint32_t count, F fn) {
  const int n = g_nctx.load();
  std::vector<uint32_t> owner(count);
  int rc = zes_partition(sizes, count, (uint32_t)n, owner.data());
  if (rc) return rc;
  std::vector<std::vector<uint32_t>> ids((size_t)n);
  for (uint32_t i = 0; i < count; i++) ids[owner[i]].push_back(i);
  std::vector<int> rcs((size_t)n, ZES_OK);
  std::vector<std::thread> th;
  for (int d = 1; d < n; d++)
    if (!ids[(size_t)d].empty()) th.emplace_back([&, d] {
      UseDev ud(d);
      rcs[(size_t)d] = fn(ids[(size_t)d]);
    });
  if (!ids[0].empty()) {
    UseDev ud(0);
    rcs[0] = fn(ids[0]);
  }
  for (auto& t : th) t.join();
  for (int r : rcs)
    if (r) return r;
  return ZES_OK;
}
extern "C" {

int zes_deflate_batch(const uint8_t* const* in, const uint64_t* in_len, uint8_t* const* out, const uint64_t* out_cap,
                      uint64_t* out_len, int32_t* status, uint32_t count) {
  if (!in || !in_len || !out || !out_cap || !out_len || !status) return ZES_E_ARG;
  if (g_nctx.load() <= 1 || count <= 1 || t_routed) {
    UseDev ud(route_host());
    return deflate_batch_one(in, in_len, out, out_cap, out_len, status, count);
  }
  return over_devices(in_len, count, [&](const std::vector<uint32_t>& ids) {
    const uint32_t m = (uint32_t)ids.size();
    std::vector<const uint8_t*> sin(m);
    std::vector<uint8_t*> sout(m);
    std::vector<uint64_t> slen(m), scap(m), sol(m);
    std::vector<int32_t> sst(m);
    for (uint32_t k = 0; k < m; k++) {
      sin[k] = in[ids[k]];
      slen[k] = in_len[ids[k]];
      sout[k] = out[ids[k]];
      scap[k] = out_cap[ids[k]];
    }
    const int rc = deflate_batch_one(sin.data(), slen.data(), sout.data(), scap.data(), sol.data(), sst.data(), m);
    if (rc) return rc;
    for (uint32_t k = 0; k < m; k++) {
      out_len[ids[k]] = sol[k];
      status[ids[k]] = sst[k];
    }
    return (int)ZES_OK;
  });
}

struct SubAlloc {  // a share's buffer k is the caller's buffer ids[k]
  zes_alloc_fn fn;
  void* user;
  const uint32_t* ids;
};
static uint8_t* sub_alloc(void* u, uint32_t k, uint64_t n) {
  const SubAlloc* a = static_cast<const SubAlloc*>(u);
  return a->fn(a->user, a->ids[k], n);
}

int zes_inflate_batch_alloc(const uint8_t* const* in, const uint64_t* in_len, zes_alloc_fn alloc, void* user, uint64_t* out_len,
                            int32_t* status, uint32_t count, uint32_t flags) {
  if (!in || !in_len || !alloc || !out_len || !status) return ZES_E_ARG;
  if (g_nctx.load() <= 1 || count <= 1 || t_routed) {
    UseDev ud(route_host());
    return inflate_batch_alloc_one(in, in_len, alloc, user, out_len, status, count, flags);
  }
  return over_devices(in_len, count, [&](const std::vector<uint32_t>& ids) {
    const uint32_t m = (uint32_t)ids.size();
    std::vector<const uint8_t*> sin(m);
    std::vector<uint64_t> slen(m), sol(m);
    std::vector<int32_t> sst(m);
    for (uint32_t k = 0; k < m; k++) {
      sin[k] = in[ids[k]];
      slen[k] = in_len[ids[k]];
    }
    SubAlloc sa{alloc, user, ids.data()};
    const int rc = inflate_batch_alloc_one(sin.data(), slen.data(), sub_alloc, &sa, sol.data(), sst.data(), m, flags);
    if (rc) return rc;
    for (uint32_t k = 0; k < m; k++) {
      out_len[ids[k]] = sol[k];
      status[ids[k]] = sst[k];
    }
    return (int)ZES_OK;
  });
}

static int adler32_locked(const uint8_t* d_in, uint64_t n, uint32_t* adler_out) {
  int rc;
  if ((rc = ensure(g.adler, 16))) return rc;
  unsigned long long* acc = (unsigned long long*)g.adler.p;
  HIPCHK(hipMemsetAsync(acc, 0, 16, g.stream));
  if (n) {
    Timed t("k_adler");
    const uint32_t nch = (uint32_t)((n + ADLER_CHUNK - 1) / ADLER_CHUNK);
    hipLaunchKernelGGL(k_adler, dim3(nch), dim3(ADLER_THREADS), 0, g.stream, d_in, (uint64_t)0, n, acc);
  }
  HIPCHK(hipMemcpyAsync(g.pinned, acc, 16, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  collect_times();
  const unsigned long long* h = (const unsigned long long*)g.pinned;
  const uint32_t s1 = (uint32_t)((1ull + h[0]) % 65521ull);
  const uint32_t s2 = (uint32_t)((n % 65521ull + h[1]) % 65521ull);
  *adler_out = (s2 << 16) | s1;
  return ZES_OK;
}

// ---- one buffer over several GPUs: block ranges and their join ----
int zes_deflate_range_dev(const uint8_t* d_in, uint64_t n, uint64_t n_readable, int final_range, uint8_t* d_out, uint64_t cap,
                          uint64_t* out_bits, uint32_t* adler) {
  ROUTE_DEV(d_in, d_out);
  if (!out_bits || !d_in || !d_out || n == 0 || n_readable < n) return ZES_E_ARG;
  if (!final_range && (n % ZES_BLK)) return ZES_E_ARG;  // only the input's last range may end inside a block
  if ((n % ZES_BLK) == 1) return ZES_E_CORRUPT;         // the reference throws on a 1-byte last block
  if ((((uintptr_t)d_in) & 15u) || (((uintptr_t)d_out) & 15u)) return ZES_E_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  uint64_t zero = 0, bits = 0;
  int32_t st = 0;
  uint32_t fl = ZES_BUF_RANGE | (final_range ? 0u : ZES_BUF_NOTFINAL), ad = 1;
  rc = deflate_batch_core(d_in, &zero, &n, d_out, &zero, &cap, &bits, &st, 1, &n_readable, &fl, &ad);
  if (rc) return rc;
  if (st) {
    if (st == ZES_E_NOSPACE) *out_bits = bits;  // (the capacity needed, in bytes)
    return st;
  }
  *out_bits = bits;
  if (adler) *adler = ad;
  return ZES_OK;
}

int zes_deflate_join_dev(const uint8_t* const* d_piece, const uint64_t* piece_bits, const uint32_t* piece_adler, const uint64_t* piece_len,
                         uint32_t count, uint8_t* d_out, uint64_t cap, uint64_t* out_len) {
  ROUTE_DEV(d_out, (d_piece && count) ? d_piece[0] : nullptr);
  if (!d_piece || !piece_bits || !piece_adler || !piece_len || !d_out || !out_len || count == 0) return ZES_E_ARG;
  if (((uintptr_t)d_out) & 15u) return ZES_E_ARG;
  uint64_t bits = 0;
  // Adler-32 of the concatenation (src/adler32.ts:1-10): s1 adds up; a piece's s2 also sees len * (s1 so far - 1)
  uint64_t s1 = 1, s2 = 0;
  for (uint32_t i = 0; i < count; i++) {
    if (piece_bits[i] && (!d_piece[i] || (((uintptr_t)d_piece[i]) & 3u))) return ZES_E_ARG;
    bits += piece_bits[i];
    const uint64_t a1 = piece_adler[i] & 0xFFFFu, a2 = piece_adler[i] >> 16;
    s2 = (s2 + a2 + (piece_len[i] % 65521u) * ((s1 + 65520u) % 65521u)) % 65521u;
    s1 = (s1 + a1 + 65520u) % 65521u;
  }
  const uint64_t raw_end = 2 + (bits + 7) / 8, total = raw_end + 4;
  *out_len = total;
  // the pieces are placed dword by dword: the kernel writes (and the memset clears) up to the dword that holds the
  // result's last byte, so the buffer must reach that far (include/zes.h says so; zes_deflate_bound always does)
  if (((total + 3) & ~3ull) > cap) return ZES_E_NOSPACE;
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(d_out, 0, (total + 3) & ~3ull, g.stream));
  uint64_t pos = 16;  // behind 78 9C
  for (uint32_t i = 0; i < count; i++) {
    if (!piece_bits[i]) continue;
    Timed t("k_bits_place");
    const uint64_t dwords = (piece_bits[i] + 63) / 32;
    const uint32_t nwg = (uint32_t)std::min<uint64_t>((dwords + 255) / 256, 4096);
    hipLaunchKernelGGL(k_bits_place, dim3(nwg), dim3(256), 0, g.stream, (uint32_t*)d_out, pos, (const uint32_t*)d_piece[i], piece_bits[i]);
    pos += piece_bits[i];
  }
  HIPCHK(hipGetLastError());
  uint8_t* hp = (uint8_t*)g.pinned;
  hp[0] = 0x78;  // src/zlib.ts:29-34
  hp[1] = 0x9C;
  const uint32_t ad = (uint32_t)((s2 << 16) | s1);
  for (int k = 0; k < 4; k++) hp[8 + k] = (uint8_t)(ad >> (24 - 8 * k));  // big-endian trailer (src/zlib.ts:37-40)
  HIPCHK(hipMemcpyAsync(d_out, hp, 2, hipMemcpyHostToDevice, g.stream));
  HIPCHK(hipMemcpyAsync(d_out + raw_end, hp + 8, 4, hipMemcpyHostToDevice, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  collect_times();
  return ZES_OK;
}

int zes_inflate_range_dev(const uint8_t* d_in, uint64_t c, uint64_t lo_bit, uint64_t own_bit, int exact_start, uint8_t* d_out, uint64_t cap,
                          uint64_t* out_len, uint64_t* first_bit, uint64_t* end_bit, uint32_t* nblocks, int* final_block) {
  ROUTE_DEV(d_in, d_out);
  if (!d_in || !out_len || !first_bit || !end_bit || !nblocks || !final_block || lo_bit < 16 || own_bit <= lo_bit) return ZES_E_ARG;
  if ((((uintptr_t)d_in) & 15u) || (((uintptr_t)d_out) & 15u) || c >= (1ull << 29)) return ZES_E_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  RangeRes rr;
  if ((rc = inflate_t1_range(d_in, 0, c, lo_bit, own_bit, exact_start != 0, d_out, 0, cap, ZES_F_DEFAULT, &rr))) return rc;
  collect_times();
  if (!rr.handled) return ZES_E_NOTRANGE;
  *out_len = rr.out_len;
  *first_bit = rr.first_bit;
  *end_bit = rr.end_bit;
  *nblocks = rr.nblocks;
  *final_block = rr.final_block ? 1 : 0;
  return rr.out_len > cap ? ZES_E_NOSPACE : ZES_OK;
}

int zes_adler32_dev(const uint8_t* d_in, uint64_t n, uint32_t* adler_out) {
  ROUTE_DEV(d_in);
  if (!adler_out) return ZES_E_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  return adler32_locked(d_in, n, adler_out);
}

// ---- raw DEFLATE (src/deflate.ts:14, src/inflate.ts:16): thin forms over the wrapped pipeline ----
// The raw stream is decoded as the body of a zlib stream whose two header bytes are supplied here; it is
// copied device-to-device behind them so that the kernels keep their aligned dword view of the input.
static int inflate_raw_staged(uint64_t n, uint8_t* d_out, uint64_t cap, uint64_t* out_len, uint32_t flags) {
  HIPCHK(hipMemsetAsync(g.st_in.p, 0x78, 1, g.stream));
  HIPCHK(hipMemsetAsync((uint8_t*)g.st_in.p + 1, 0x9C, 1, g.stream));
  return inflate_one((const uint8_t*)g.st_in.p, 0, n + 2, d_out, 0, cap, out_len, flags, 0x78);
}

int zes_inflate_raw_dev(const uint8_t* d_in, uint64_t c, uint64_t offset, uint8_t* d_out, uint64_t cap, uint64_t* out_len,
                        uint32_t flags) {
  ROUTE_DEV(d_in, d_out);
  if (!out_len || (!d_in && c)) return ZES_E_ARG;
  if ((((uintptr_t)d_out) & 15u)) return ZES_E_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  const uint64_t n = offset < c ? c - offset : 0;
  if ((rc = ensure(g.st_in, n + 2 + 64))) return rc;
  if (n) HIPCHK(hipMemcpyAsync((uint8_t*)g.st_in.p + 2, d_in + offset, n, hipMemcpyDeviceToDevice, g.stream));
  return inflate_raw_staged(n, d_out, cap, out_len, flags);
}

int zes_inflate_raw(const uint8_t* in, uint64_t c, uint64_t offset, uint8_t* out, uint64_t cap, uint64_t* out_len, uint32_t flags) {
  UseDev ud(route_host());
  if (!out_len || (!in && c)) return ZES_E_ARG;
  *out_len = 0;
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  const uint64_t n = offset < c ? c - offset : 0;
  if ((rc = ensure(g.st_in, n + 2 + 64))) return rc;
  if ((rc = upload((uint8_t*)g.st_in.p + 2, in + offset, n))) return rc;
  // decode into pooled device memory: grow-and-retry like the reference's Uint8WriteStream
  uint64_t dcap = std::max<uint64_t>(cap, std::max<uint64_t>(n * 4, 1 << 20));
  for (int attempt = 0; attempt < 8; attempt++) {
    if ((rc = ensure(g.st_out, dcap + 64))) return rc;
    uint64_t m = 0;
    rc = inflate_raw_staged(n, (uint8_t*)g.st_out.p, dcap, &m, flags);
    if (rc == ZES_E_NOSPACE && m > dcap) {
      dcap = m;
      continue;
    }
    if (rc) return rc;
    *out_len = m;
    if (m > cap) return ZES_E_NOSPACE;
    return download(out, (const uint8_t*)g.st_out.p, m);
  }
  return ZES_E_DEVICE;
}

static int deflate_raw_common(const uint8_t* d_in, uint64_t n, uint64_t* raw_len) {
  const uint64_t bound = deflate_bound(n);
  int rc;
  if ((rc = ensure(g.st_out, bound + 64))) return rc;
  uint64_t zero = 0, dl = 0;
  int32_t st = 0;
  rc = deflate_batch_core(d_in, &zero, &n, (uint8_t*)g.st_out.p, &zero, &bound, &dl, &st, 1);
  if (rc) return rc;
  if (st) return st;
  *raw_len = dl - 6;  // without 78 9C and the Adler-32 trailer (src/zlib.ts:28-46)
  return ZES_OK;
}

int zes_deflate_raw_dev(const uint8_t* d_in, uint64_t n, uint8_t* d_out, uint64_t cap, uint64_t* out_len) {
  ROUTE_DEV(d_in, d_out);
  if (!out_len || !d_out) return ZES_E_ARG;
  *out_len = 0;
  if (deflate_throws(n)) return ZES_E_CORRUPT;
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  uint64_t rl = 0;
  if ((rc = deflate_raw_common(d_in, n, &rl))) return rc;
  *out_len = rl;
  if (rl > cap) return ZES_E_NOSPACE;
  HIPCHK(hipMemcpyAsync(d_out, (const uint8_t*)g.st_out.p + 2, rl, hipMemcpyDeviceToDevice, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  return ZES_OK;
}

int zes_deflate_raw(const uint8_t* in, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* out_len) {
  UseDev ud(route_host());
  if (!out_len || (!in && n) || !out) return ZES_E_ARG;
  *out_len = 0;
  if (deflate_throws(n)) return ZES_E_CORRUPT;
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  if ((rc = ensure(g.st_in, n + 64))) return rc;
  if ((rc = upload((uint8_t*)g.st_in.p, in, n))) return rc;
  uint64_t rl = 0;
  if ((rc = deflate_raw_common((const uint8_t*)g.st_in.p, n, &rl))) return rc;
  *out_len = rl;
  if (rl > cap) return ZES_E_NOSPACE;
  return download(out, (const uint8_t*)g.st_out.p + 2, rl);
}

int zes_adler32(const uint8_t* in, uint64_t n, uint32_t* adler_out) {
  UseDev ud(route_host());
  if (!adler_out || (!in && n)) return ZES_E_ARG;
  std::lock_guard<std::mutex> lk(g_mu);  // staging and kernel under one lock: nobody else's call can replace st_in in between
  int rc = init_locked(-1);
  if (rc) return rc;
  if ((rc = ensure(g.st_in, n + 64))) return rc;
  if ((rc = upload((uint8_t*)g.st_in.p, in, n))) return rc;
  return adler32_locked((const uint8_t*)g.st_in.p, n, adler_out);
}

int zes_stage_lz77_dev(const uint8_t* d_in, uint64_t n, uint64_t start, uint32_t len, uint32_t* h_tokens, uint32_t* ntokens) {
  ROUTE_DEV(d_in);
  if (!h_tokens || !ntokens || len < 2 || len > ZES_BLK || start + len > n || (start % ZES_BLK)) return ZES_E_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  ZesBuf b;
  memset(&b, 0, sizeof b);
  b.in_off = 0;
  b.n = n;  // the halo reads up to the real input end
  b.n_read = n;
  b.nblk = (uint32_t)((n + ZES_BLK - 1) / ZES_BLK);
  ZesBlk z;
  memset(&z, 0, sizeof z);
  z.buf = 0;
  z.blk = (uint32_t)(start / ZES_BLK);
  z.len = len;
  if ((rc = ensure(g.bufs, sizeof b))) return rc;
  if ((rc = ensure(g.blks, sizeof z))) return rc;
  if ((rc = ensure(g.idx_a, (size_t)ZES_BLK * 4))) return rc;
  if ((rc = ensure(g.idx_b, (size_t)ZES_BLK * 4))) return rc;
  if ((rc = ensure(g.sdelta, (size_t)ZES_BLK * 2 + 64))) return rc;
  if ((rc = ensure(g.hists, 320 * 4))) return rc;
  if ((rc = ensure(g.tmask, ZES_TMASK_WORDS * 4))) return rc;
  if ((rc = ensure(g.mlist, ZES_MLIST_WORDS * 4))) return rc;
  HIPCHK(hipMemcpyAsync(g.bufs.p, &b, sizeof b, hipMemcpyHostToDevice, g.stream));
  HIPCHK(hipMemcpyAsync(g.blks.p, &z, sizeof z, hipMemcpyHostToDevice, g.stream));
  const bool use_index = getenv("ZES_NO_INDEX") == nullptr;  // the same three launches as the whole pipeline
  hipLaunchKernelGGL(k_lz_sort, dim3(1), dim3(SORT_THREADS), 0, g.stream, d_in, (const ZesBuf*)g.bufs.p, (const ZesBlk*)g.blks.p,
                     (uint32_t*)g.idx_a.p, (uint32_t*)g.idx_b.p, (uint32_t*)g.idx_a.p, (uint16_t*)g.sdelta.p,
                     ZES_SORT_MODE_FIRST | (use_index ? ZES_SORT_USE_INDEX : 0u));
  if (use_index) {
    hipLaunchKernelGGL(k_lz_index, dim3(1), dim3(IDX_THREADS), 0, g.stream, d_in, (const ZesBuf*)g.bufs.p, (const ZesBlk*)g.blks.p,
                       (uint32_t*)g.idx_a.p, (uint32_t*)g.idx_b.p, (uint32_t*)g.idx_a.p, (uint16_t*)g.sdelta.p);
    hipLaunchKernelGGL(k_lz_sort, dim3(1), dim3(SORT_THREADS), 0, g.stream, d_in, (const ZesBuf*)g.bufs.p, (const ZesBlk*)g.blks.p,
                       (uint32_t*)g.idx_a.p, (uint32_t*)g.idx_b.p, (uint32_t*)g.idx_a.p, (uint16_t*)g.sdelta.p, ZES_SORT_MODE_REDO);
  }
  if (const char* dump = getenv("ZES_DUMP_INDEX")) {  // development: the block's index as the match finders will see it
    HIPCHK(hipStreamSynchronize(g.stream));
    std::vector<uint32_t> hinv(ZES_BLK), hflag(1);
    std::vector<uint16_t> hsd(ZES_BLK);
    HIPCHK(hipMemcpy(hinv.data(), g.idx_a.p, ZES_BLK * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hsd.data(), g.sdelta.p, ZES_BLK * 2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hflag.data(), (uint32_t*)g.idx_a.p + ZES_BLK - 1, 4, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(dump, "wb")) {
      fwrite(hflag.data(), 4, 1, f);
      fwrite(hinv.data(), 4, ZES_BLK, f);
      fwrite(hsd.data(), 2, ZES_BLK, f);
      fclose(f);
    }
  }
  hipLaunchKernelGGL(k_lz_match, dim3(1), dim3(MATCH_THREADS), 0, g.stream, d_in, (const ZesBuf*)g.bufs.p,
                     (const ZesBlk*)g.blks.p, (const uint32_t*)g.idx_a.p, (uint32_t*)g.idx_b.p, (uint32_t*)g.mlist.p);
  hipLaunchKernelGGL(k_lz_match_lazy, dim3(1), dim3(MATCH_THREADS), 0, g.stream, d_in, (const ZesBuf*)g.bufs.p,
                     (const ZesBlk*)g.blks.p, (const uint32_t*)g.idx_a.p, (const uint32_t*)g.idx_a.p, (const uint16_t*)g.sdelta.p, (uint32_t*)g.idx_b.p,
                     (uint32_t*)g.tmask.p, (uint32_t*)g.mlist.p, (const uint32_t*)nullptr);
  hipLaunchKernelGGL(k_lz_parse_small, dim3(1), dim3(PARSE_THREADS), 0, g.stream, d_in, (const ZesBuf*)g.bufs.p, (ZesBlk*)g.blks.p,
                     (const uint32_t*)g.idx_b.p, (uint32_t*)g.idx_a.p, (uint32_t*)g.hists.p, (const uint32_t*)g.tmask.p, (const uint32_t*)g.mlist.p);
  hipLaunchKernelGGL(k_lz_parse, dim3(1), dim3(PARSE_THREADS), 0, g.stream, d_in, (const ZesBuf*)g.bufs.p, (ZesBlk*)g.blks.p,
                     (const uint32_t*)g.idx_b.p, (uint32_t*)g.idx_a.p, (uint32_t*)g.hists.p, (const uint32_t*)g.tmask.p, (const uint32_t*)g.mlist.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(&z, g.blks.p, sizeof z, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  *ntokens = z.ntok;
  HIPCHK(hipMemcpy(h_tokens, g.idx_a.p, (size_t)z.ntok * 4, hipMemcpyDeviceToHost));
  return ZES_OK;
}

int zes_stage_huff_lengths_dev(const uint32_t* h_hist, uint32_t nsym, uint32_t maxlen, uint8_t* h_lens) {
  if (!h_hist || !h_lens || nsym == 0 || nsym > 288 || maxlen == 0 || maxlen > 15) return ZES_E_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  if ((rc = ensure(g.hists, 320 * 4))) return rc;
  if ((rc = ensure(g.codes, 320))) return rc;
  HIPCHK(hipMemcpyAsync(g.hists.p, h_hist, nsym * 4, hipMemcpyHostToDevice, g.stream));
  hipLaunchKernelGGL(k_huff_lengths_only, dim3(1), dim3(HUFF_THREADS_HOST), 0, g.stream, (const uint32_t*)g.hists.p, nsym, maxlen,
                     (uint8_t*)g.codes.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(h_lens, g.codes.p, nsym, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  return ZES_OK;
}

int zes_selftest_lds_order(uint32_t iters, uint32_t seed, uint64_t* bad, uint64_t* checked) {
  if (!bad || !checked || iters == 0 || iters > 100000u) return ZES_E_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = init_locked(-1);
  if (rc) return rc;
  if ((rc = ensure(g.hists, 320 * 4))) return rc;
  HIPCHK(hipMemsetAsync(g.hists.p, 0, 16, g.stream));
  hipLaunchKernelGGL(k_selftest_lds_order, dim3(256), dim3(SORT_THREADS), 0, g.stream, (unsigned long long*)g.hists.p, iters, seed);
  HIPCHK(hipGetLastError());
  unsigned long long h[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(h, g.hists.p, 16, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  *bad = h[0];
  *checked = h[1];
  return ZES_OK;
}

int zes_last_inflate_tier(void) {
  UseDev ud(t_last);  // the context that served this thread's last call (single host calls go round robin over the contexts)
  std::lock_guard<std::mutex> lk(g_mu);
  return g.last_tier;
}

int zes_set_profiling(int on) {
  std::lock_guard<std::mutex> cfg(g_cfg_mu);
  for (int i = 0; i < ZES_MAX_DEV; i++) {  // every context: a call may be served by any of them
    std::lock_guard<std::mutex> lk(g_mus[i]);
    g_ctx[i].profiling = on != 0;
  }
  return ZES_OK;
}

uint64_t zes_pool_bytes(void) {
  uint64_t total = 0;
  for (int i = 0; i < ZES_MAX_DEV; i++) {
    std::lock_guard<std::mutex> lk(g_mus[i]);
    Ctx& c = g_ctx[i];
    const DevBuf* all[] = {&c.bufs, &c.blks, &c.idx_a, &c.idx_b, &c.sdelta, &c.tmask, &c.mlist, &c.hists, &c.codes, &c.hdrs, &c.adler, &c.res, &c.order, &c.surv, &c.vlong, &c.segfail, &c.symoff, &c.cand,
                           &c.cand_sorted, &c.counters, &c.cres, &c.map, &c.resume, &c.dbg, &c.ibufs, &c.ibufs2, &c.mvlist, &c.scratch, &c.st_in, &c.st_out,
                           &c.sres, &c.maps, &c.seglist, &c.segprefix, &c.wins, &c.sym16, &c.segorder, &c.segjobs, &c.pw16, &c.gwins, &c.seglive, &c.segouts};
    for (const DevBuf* b : all) total += b->cap;
  }
  return total;
}

int zes_last_kernel_times(zes_ktime* out, int cap) {
  UseDev ud(t_last);
  std::lock_guard<std::mutex> lk(g_mu);
  int n = 0;
  g.name_pool.clear();
  for (auto& e : g.last_times) g.name_pool.push_back(e.first);
  for (size_t i = 0; i < g.last_times.size() && n < cap; i++, n++) {
    out[n].name = g.name_pool[i].c_str();
    out[n].ms = (float)g.last_times[i].second.ms;
    out[n].launches = g.last_times[i].second.launches;
  }
  return n;
}

}  // extern "C"
