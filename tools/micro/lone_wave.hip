// Microbenchmark: what single instructions cost a lone wavefront on gfx950 (cycles per loop round, s_memtime).
// build: hipcc -O3 --offload-arch=gfx950 lone_wave.hip -o lone_wave ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 4096
__global__ void k(uint64_t* out, uint32_t seed) {
  __shared__ uint32_t lds[2048];
  const uint32_t lane = threadIdx.x;
  for (uint32_t i = lane; i < 2048; i += 64) lds[i] = (i * 2654435761u + seed) & 2047u;
  uint32_t tab[32];
#pragma unroll
  for (int r = 0; r < 32; r++) tab[r] = lds[r * 64 + lane];
  __syncthreads();
  uint64_t t0, t1;
  uint32_t x = seed & 2047u, acc = 0;
  // (a) dependent scalar adds/shifts: 8 per round
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) {
    x = (x * 5u + 1u) & 2047u; x ^= x >> 3; x = (x + 7u) & 2047u; x ^= x >> 2;
    asm volatile("" : "+s"(x));
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[0] = t1 - t0;
  acc += x;
  // (b) LDS lookup -> readfirstlane, dependent chain
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) x = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds[x]);
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[1] = t1 - t0;
  acc += x;
  // (c) VGPR table lookup: indexed register move + readlane, dependent chain
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) x = (uint32_t)__builtin_amdgcn_readlane((int)tab[x >> 6], (int)(x & 63u));
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[2] = t1 - t0;
  acc += x;
  // (d) a taken uniform branch per round (two-way, alternating)
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) {
    if (x & 1u) { x = x * 3u + 1u; asm volatile("" : "+s"(x)); } else { x = x + 5u; asm volatile("" : "+s"(x)); }
    x &= 2047u;
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[3] = t1 - t0;
  acc += x;
  // (e) LDS byte store by lane 0 + LDS read/write copy (64 lanes), dependent through memory order only
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) {
    if (lane == 0) reinterpret_cast<volatile uint8_t*>(lds)[(i * 7) & 4095] = (uint8_t)i;
    asm volatile("");
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[4] = t1 - t0;
  // (e2) the same store without 'volatile' (no wait for it), (e3) by all lanes (same address, same value)
  uint8_t* l8 = reinterpret_cast<uint8_t*>(lds);
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) {
    if (lane == 0) l8[(i * 7) & 4095] = (uint8_t)i;
    asm volatile("" ::: "memory");
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[8] = t1 - t0;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) {
    l8[(i * 7) & 4095] = (uint8_t)i;
    asm volatile("" ::: "memory");
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[9] = t1 - t0;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) {
    const uint32_t s = (i * 13 + lane) & 2047u, dd = (i * 29 + 64 + lane) & 2047u;
    lds[dd] = lds[s];
    asm volatile("");
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[5] = t1 - t0;
  // (f) 8 independent VALU ops per round
  uint32_t v = lane;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) {
    v = v * 5u + 1u; v ^= v >> 3; v += 7u; v ^= v >> 2; v = v * 3u; v ^= v >> 5; v += 11u; v ^= v >> 1;
    asm volatile("" : "+v"(v));
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[6] = t1 - t0;
  if (lane == 0) out[7] = acc + v + lds[5];
}
int main() {
  uint64_t* d; hipMalloc(&d, 128);
  uint64_t h[16];
  for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 12345u + rep); hipDeviceSynchronize(); }
  hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
  const char* names[7] = {"8 dependent scalar ops", "LDS read + readfirstlane (dependent)", "VGPR-table lookup (gpr_idx + readlane)", "uniform two-way branch + 2 scalar ops",
                          "lane-0 LDS byte store", "64-lane LDS copy (read, write)", "8 dependent vector ops"};
  printf("%-42s %7.1f cycles per round\n", "lane-0 LDS byte store, not volatile", (double)h[8] / N);
  printf("%-42s %7.1f cycles per round\n", "all-lane LDS byte store, same address", (double)h[9] / N);
  for (int i = 0; i < 7; i++) printf("%-42s %7.1f cycles per round\n", names[i], (double)h[i] / N);
  return 0;
}
