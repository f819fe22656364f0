// zes_napi.cc — N-API addon: the binding between the TypeScript façade (zlib.ts) and the C-ABI of
// include/zes.h.  Thin on purpose: argument marshalling and error translation only; every byte
// of work happens in libzes_hip.so (HIP kernels).  Synchronous like the reference's functions
// (src/zlib.ts:11,25); deflateAsync / inflateAsync run the same calls on the libuv thread pool and return
// Promises (the JS thread stays free while a GPU works).  N-API version 3 calls only (Node >= 10).
#include <node_api.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>

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

void free_external(napi_env, void* data, void*);
// the library's output buffer becomes the result's ArrayBuffer (shrunk to the exact length: `buffer.byteLength === length`
// like src/zlib.ts:42): no second copy of the result on the JS thread.  Takes the buffer over (frees it on failure).
napi_value take_u8(napi_env env, uint8_t* buf, size_t n) {
  void* shrunk = realloc(buf, n ? n : 1);
  if (shrunk) buf = static_cast<uint8_t*>(shrunk);
  napi_value ab, ta;
  if (napi_create_external_arraybuffer(env, buf, n, free_external, nullptr, &ab) != napi_ok) {
    free(buf);
    return nullptr;
  }
  if (napi_create_typedarray(env, napi_uint8_array, n, ab, 0, &ta) != napi_ok) return nullptr;  // (the ArrayBuffer's finalizer owns buf)
  return ta;
}

// zes_alloc_fn of the synchronous inflate: runs on the JS thread, inside zes_inflate_alloc
struct SyncAlloc {
  napi_env env;
  napi_value ab;
  bool failed;
};
uint8_t* sync_alloc(void* user, uint32_t, uint64_t n) {
  SyncAlloc* a = static_cast<SyncAlloc*>(user);
  void* dst = nullptr;
  if (napi_create_arraybuffer(a->env, (size_t)n, &dst, &a->ab) != napi_ok) {
    a->failed = true;
    return nullptr;
  }
  static uint8_t empty;
  return n ? static_cast<uint8_t*>(dst) : &empty;
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
  uint8_t* tmp = static_cast<uint8_t*>(malloc(cap));
  if (!tmp) return throw_status(env, ZES_E_ARG);
  const int rc = zes_deflate(in, n, tmp, cap, &out_len);
  if (rc) {
    free(tmp);
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
  SyncAlloc sa{env, nullptr, false};
  uint64_t out_len = 0;
  const int rc = zes_inflate_alloc(in, c, sync_alloc, &sa, &out_len, ZES_F_DEFAULT);
  if (rc) return sa.failed ? nullptr : throw_status(env, rc);  // (a failed napi allocation has its own pending exception)
  napi_value ta;
  if (napi_create_typedarray(env, napi_uint8_array, out_len, sa.ab, 0, &ta) != napi_ok) return nullptr;
  return ta;
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
  uint8_t* tmp = static_cast<uint8_t*>(malloc(cap));
  if (!tmp) return throw_status(env, ZES_E_ARG);
  const int rc = zes_deflate_raw(in, n, tmp, cap, &out_len);
  if (rc) {
    free(tmp);
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
    uint8_t* tmp = static_cast<uint8_t*>(malloc(cap));
    if (!tmp) return throw_status(env, ZES_E_ARG);
    const int rc = zes_inflate_raw(in, c, offset, tmp, cap, &out_len, ZES_F_DEFAULT);
    if (rc == ZES_E_NOSPACE && out_len > cap) {
      free(tmp);
      cap = out_len;
      continue;
    }
    if (rc) {
      free(tmp);
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
  uint8_t* out = nullptr;
  uint64_t out_len = 0;
  int rc = 0;
};

void async_execute(napi_env, void* data) {  // worker thread: no N-API calls here
  AsyncJob* j = static_cast<AsyncJob*>(data);
  if (!j->inflate) {
    uint64_t cap = 0;
    zes_deflate_bound(j->n, &cap);
    j->out = static_cast<uint8_t*>(malloc(cap ? cap : 1));
    j->rc = j->out ? zes_deflate(j->in, j->n, j->out, cap, &j->out_len) : ZES_E_ARG;
    return;
  }
  // grow-and-retry (the size/fetch pair of the synchronous form keeps state in the library between two calls,
  // which another thread's call could replace)
  uint64_t cap = (uint64_t)j->n * 4 + 65536;
  for (int attempt = 0; attempt < 8; attempt++) {
    j->out = static_cast<uint8_t*>(malloc(cap));
    if (!j->out) {
      j->rc = ZES_E_ARG;
      return;
    }
    j->rc = zes_inflate(j->in, j->n, j->out, cap, &j->out_len, ZES_F_DEFAULT);
    if (j->rc == ZES_E_NOSPACE && j->out_len > cap) {
      free(j->out);
      j->out = nullptr;
      cap = j->out_len;
      continue;
    }
    return;
  }
  j->rc = ZES_E_DEVICE;
}

void free_external(napi_env, void* data, void*) { free(data); }

void async_complete(napi_env env, napi_status, void* data) {  // JS thread again
  AsyncJob* j = static_cast<AsyncJob*>(data);
  napi_value result = nullptr;
  bool ok = j->rc == 0;
  if (ok) {
    napi_value ab;
    // the worker's buffer becomes the result's ArrayBuffer (exact length, no copy on the JS thread)
    if (napi_create_external_arraybuffer(env, j->out, (size_t)j->out_len, free_external, nullptr, &ab) == napi_ok &&
        napi_create_typedarray(env, napi_uint8_array, (size_t)j->out_len, ab, 0, &result) == napi_ok) {
      j->out = nullptr;  // owned by the ArrayBuffer now
    } else {
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
  free(j->out);
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
  uint64_t* out_cap = nullptr;
  uint64_t* out_len = nullptr;
  int32_t* status = nullptr;
  int rc = 0;
  ~BatchJob() {
    if (out)
      for (uint32_t i = 0; i < count; i++) free(out[i]);
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
  j->out[i] = static_cast<uint8_t*>(malloc(n ? n : 1));
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
    j->out[i] = static_cast<uint8_t*>(malloc(j->out_cap[i]));
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
      napi_value ab;
      void* shrunk = realloc(j->out[i], j->out_len[i] ? (size_t)j->out_len[i] : 1);  // exact-size backing store
      if (shrunk) j->out[i] = static_cast<uint8_t*>(shrunk);
      if (napi_create_external_arraybuffer(env, j->out[i], (size_t)j->out_len[i], free_external, nullptr, &ab) != napi_ok) return nullptr;
      j->out[i] = nullptr;  // owned by the ArrayBuffer now
      if (napi_create_typedarray(env, napi_uint8_array, (size_t)j->out_len[i], ab, 0, &v) != napi_ok) return nullptr;
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
void free_pinned(napi_env, void* data, void*) { zes_host_free(data); }
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
  napi_value ab, ta;
  if (napi_create_external_arraybuffer(env, p, (size_t)d, free_pinned, nullptr, &ab) != napi_ok) {
    zes_host_free(p);  // no ArrayBuffer took the block over
    return nullptr;
  }
  // from here on the ArrayBuffer's finalizer (free_pinned) owns the block, whatever happens to the view
  if (napi_create_typedarray(env, napi_uint8_array, (size_t)d, ab, 0, &ta) != napi_ok) return nullptr;
  return ta;
}

napi_value ModuleInit(napi_env env, napi_value exports) {
  napi_property_descriptor props[] = {
      {"deflate", nullptr, Deflate, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"inflate", nullptr, Inflate, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"deflateRaw", nullptr, DeflateRaw, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"inflateRaw", nullptr, InflateRaw, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"deflateAsync", nullptr, DeflateAsync, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"inflateAsync", nullptr, InflateAsync, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"deflateBatch", nullptr, DeflateBatch, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"inflateBatch", nullptr, InflateBatch, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"deflateBatchAsync", nullptr, DeflateBatchAsync, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"inflateBatchAsync", nullptr, InflateBatchAsync, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"allocPinned", nullptr, AllocPinned, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"adler32", nullptr, Adler32, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"init", nullptr, Init, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"initDevices", nullptr, InitDevices, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"trim", nullptr, Trim, nullptr, nullptr, nullptr, napi_default, nullptr},
  };
  napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props);
  return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, ModuleInit)
