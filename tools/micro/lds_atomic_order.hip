// Microbenchmark / property check: does ds_add_rtn_u32 hand lanes of ONE wavefront that hit the same LDS word their old
// values in ascending lane order?  (k_lz_sort's digit ranks would then come from one LDS instruction instead of eight
// ballots.)  Sixteen wavefronts per workgroup, each with its own row of 128 words (two 16-bit counters per word, as the
// sort would pack them), several digit patterns; every returned value is checked against the rank computed with ballots.
// build: hipcc -O3 --offload-arch=gfx950 lds_atomic_order.hip -o lds_atomic_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__global__ __launch_bounds__(1024) void k(unsigned long long* out, uint32_t iters, uint32_t seed) {
  __shared__ uint32_t row[16][128];
  __shared__ uint32_t noise[4096];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  unsigned long long bad = 0, n = 0;
  uint64_t t = 0;
  for (uint32_t it = 0; it < iters; it++) {
    row[wave][lane] = 0;
    row[wave][lane + 64] = 0;
    const uint32_t pat = (it + wave) % 6u;
    const uint32_t h = mix(seed ^ (blockIdx.x * 0x9E3779B1u) ^ (it * 1024u + tid));
    uint32_t d;
    if (pat == 0) d = h & 255u;                     // uniform
    else if (pat == 1) d = h & 3u;                  // four digits
    else if (pat == 2) d = 77u;                     // one digit
    else if (pat == 3) d = 32u + ((h >> 3) % 27u);  // text-like
    else if (pat == 4) d = (h & 1u) + 2u * ((h >> 8) & 7u);  // pairs that share a word
    else d = (lane * 4u + (h & 3u)) & 255u;         // same bank, different words
    const bool valid = pat != 3 || (h >> 20) % 9u != 0u;  // holes
    noise[(h >> 4) & 4095u] = h;  // other LDS traffic in between
    const uint64_t t0 = __builtin_readcyclecounter();
    uint32_t rk[4];
    // four rounds back to back, like the sort: round r's digit is a rotation of the pattern
    uint32_t dd[4];
#pragma unroll
    for (int r = 0; r < 4; r++) dd[r] = (pat == 2) ? d : (d + 17u * r * (pat == 0)) & 255u;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t old = valid ? atomicAdd(&row[wave][dd[r] >> 1], 1u << (16u * (dd[r] & 1u))) : 0u;
      rk[r] = (old >> (16u * (dd[r] & 1u))) & 0xffffu;
    }
    t += __builtin_readcyclecounter() - t0;
    // expected: lanes below me with my digit in this round + all lanes with my digit in earlier rounds
    uint32_t prevcnt = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      uint64_t m = __ballot(valid);
      for (int b = 0; b < 8; b++) {
        const bool bit = (dd[r] >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
      }
      const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      // earlier rounds' lanes with this digit
      uint32_t before = 0;
      for (int q = 0; q < r; q++) {
        uint64_t mq = __ballot(valid);
        for (int b = 0; b < 8; b++) {
          const bool bit = (dd[r] >> b) & 1u;
          const uint64_t bal = __ballot((dd[q] >> b) & 1u);
          mq &= bit ? bal : ~bal;
        }
        before += (uint32_t)__popcll(mq);
      }
      if (valid) {
        n++;
        if (rk[r] != rank + before) bad++;
      }
      (void)prevcnt;
    }
  }
  atomicAdd(&out[0], bad);
  atomicAdd(&out[1], n);
  if (tid == 0 && blockIdx.x == 0) out[2] = t / iters;
  if (noise[tid] == 0x12345u) out[3] = 1;
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 64);
  hipMemset(d, 0, 64);
  for (uint32_t s = 1; s <= 8; s++) {
    hipLaunchKernelGGL(k, dim3(512), dim3(1024), 0, 0, d, 2000u, s * 7919u);
    if (hipDeviceSynchronize() != hipSuccess) {
      printf("launch failed\n");
      return 1;
    }
  }
  unsigned long long h[4];
  hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  printf("ds_add_rtn same-word order: %llu of %llu returned values differ from the lane-order rank; four atomics back to back: %llu cycles\n", h[0], h[1], h[2]);
  return h[0] != 0;
}
