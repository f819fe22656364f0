// Microbenchmark: what a scattered global store (and load) instruction costs a CU on gfx950, by address pattern.
// 256 workgroups x 1024 threads, every wave issues ROUNDS stores of the pattern into its workgroup's own 512 KiB of a
// buffer (L2-resident); cycles per instruction per CU = elapsed / (ROUNDS * 16 waves).
// build: hipcc -O3 --offload-arch=gfx950 scatter_rate.hip -o scatter_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ROUNDS 512
__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// dword index inside the workgroup's 128 Ki dwords
__device__ __forceinline__ uint32_t addr(int pat, uint32_t it, uint32_t wave, uint32_t lane) {
  const uint32_t r = mix(it * 16u + wave);
  switch (pat) {
    case 0: return ((r & 2047u) * 64u + lane) & 131071u;                                  // one run of 64 dwords
    case 1: return (mix(r + lane) & 8191u) * 16u + (lane & 15u);                         // 64 lines
    case 2: return (mix(r + (lane >> 1)) & 8191u) * 16u + (lane & 1u) + 2u * ((lane >> 1) & 7u);   // adjacent pairs
    case 3: return (mix(r + (lane >> 2)) & 8191u) * 16u + (lane & 3u) + 4u * ((lane >> 2) & 3u);   // adjacent quads
    case 4: return (mix(r + (lane >> 4)) & 8191u) * 16u + (lane & 15u);                  // four runs of 16
    case 5: return (mix(r + (lane & 15u)) & 8191u) * 16u + (lane >> 4) + 4u * (lane & 3u);  // quads, lanes 16 apart
    case 6: return (mix(r + (lane >> 3)) & 8191u) * 16u + (lane & 7u) + 8u * ((lane >> 3) & 1u);   // runs of 8
    default: return 0;
  }
}
template <int WIDTH, bool LOAD>
__global__ __launch_bounds__(1024) void k(uint32_t* buf, unsigned long long* out, int pat) {
  uint32_t* mine = buf + (size_t)blockIdx.x * 131072u * 4u;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t acc = 0;
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 4
  for (uint32_t it = 0; it < ROUNDS; it++) {
    const uint32_t a = addr(pat, it, wave, lane);
    if (WIDTH == 1) {
      if (LOAD) acc += __builtin_nontemporal_load(mine + a); else mine[a] = it;
    } else if (WIDTH == 2) {
      uint2* q = reinterpret_cast<uint2*>(mine) + a;
      if (LOAD) acc += q->x + q->y; else *q = make_uint2(it, lane);
    } else {
      uint4* q = reinterpret_cast<uint4*>(mine) + a;
      if (LOAD) acc += q->x + q->w; else *q = make_uint4(it, lane, it, lane);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (acc == 0x12345u) out[1] = 1;
}
int main() {
  uint32_t* buf;
  unsigned long long* d;
  hipMalloc(&buf, (size_t)256 * 131072 * 16);
  hipMalloc(&d, 64);
  const char* names[7] = {"one run of 64", "64 lines", "adjacent pairs", "adjacent quads", "four runs of 16", "quads, lanes 16 apart", "runs of 8"};
  for (int load = 0; load < 2; load++)
    for (int width = 1; width <= 4; width *= 2)
      for (int pat = 0; pat < 7; pat++) {
        for (int rep = 0; rep < 2; rep++) {
          if (width == 1 && !load) hipLaunchKernelGGL((k<1, false>), dim3(256), dim3(1024), 0, 0, buf, d, pat);
          if (width == 2 && !load) hipLaunchKernelGGL((k<2, false>), dim3(256), dim3(1024), 0, 0, buf, d, pat);
          if (width == 4 && !load) hipLaunchKernelGGL((k<4, false>), dim3(256), dim3(1024), 0, 0, buf, d, pat);
          if (width == 1 && load) hipLaunchKernelGGL((k<1, true>), dim3(256), dim3(1024), 0, 0, buf, d, pat);
          if (width == 2 && load) hipLaunchKernelGGL((k<2, true>), dim3(256), dim3(1024), 0, 0, buf, d, pat);
          if (width == 4 && load) hipLaunchKernelGGL((k<4, true>), dim3(256), dim3(1024), 0, 0, buf, d, pat);
          hipDeviceSynchronize();
        }
        unsigned long long h[2];
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%s %2d B/lane  %-22s %7.1f cycles per instruction per CU\n", load ? "load " : "store", 4 * width, names[pat], (double)h[0] / (ROUNDS * 16.0));
      }
  return 0;
}
