// Microbenchmark: scalar and vector instruction issue rates per CU on gfx950 with 1, 2, 4 waves per SIMD
// (independent s_add / v_add streams).  build: hipcc -O3 --offload-arch=gfx950 issue_rate.hip -o issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ROUNDS 2000
template <int KIND>
__global__ void k(uint64_t* out) {
  uint32_t a = threadIdx.x, b = 1, c = 2, d = 3, e = 4, f = 5, g = 6, h = 7;
  uint32_t sa = blockIdx.x, sb = 1, sc = 2, sd = 3, se = 4, sf = 5, sg = 6, sh = 7;
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ROUNDS; i++) {
    if (KIND == 0) {  // 32 scalar adds, 8 independent chains
      asm volatile(
          "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n"
          "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n"
          "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n"
          "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n"
          : "+s"(sa), "+s"(sb), "+s"(sc), "+s"(sd), "+s"(se), "+s"(sf), "+s"(sg), "+s"(sh)::"scc");
    } else if (KIND == 1) {  // 32 vector adds
      asm volatile(
          "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1\n"
          "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1\n"
          "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1\n"
          "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1\n"
          : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    } else {  // 16 scalar + 16 vector, interleaved
      asm volatile(
          "s_add_u32 %8, %8, 1\n v_add_u32 %0, %0, 1\n s_add_u32 %9, %9, 1\n v_add_u32 %1, %1, 1\n s_add_u32 %10, %10, 1\n v_add_u32 %2, %2, 1\n s_add_u32 %11, %11, 1\n v_add_u32 %3, %3, 1\n"
          "s_add_u32 %12, %12, 1\n v_add_u32 %4, %4, 1\n s_add_u32 %13, %13, 1\n v_add_u32 %5, %5, 1\n s_add_u32 %14, %14, 1\n v_add_u32 %6, %6, 1\n s_add_u32 %15, %15, 1\n v_add_u32 %7, %7, 1\n"
          "s_add_u32 %8, %8, 1\n v_add_u32 %0, %0, 1\n s_add_u32 %9, %9, 1\n v_add_u32 %1, %1, 1\n s_add_u32 %10, %10, 1\n v_add_u32 %2, %2, 1\n s_add_u32 %11, %11, 1\n v_add_u32 %3, %3, 1\n"
          "s_add_u32 %12, %12, 1\n v_add_u32 %4, %4, 1\n s_add_u32 %13, %13, 1\n v_add_u32 %5, %5, 1\n s_add_u32 %14, %14, 1\n v_add_u32 %6, %6, 1\n s_add_u32 %15, %15, 1\n v_add_u32 %7, %7, 1\n"
          : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+s"(sa), "+s"(sb), "+s"(sc), "+s"(sd), "+s"(se), "+s"(sf), "+s"(sg), "+s"(sh)::"scc");
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if ((a + b + c + d + e + f + g + h + sa + sb + sc + sd + se + sf + sg + sh) == 0x12345u) out[1] = 1;
}
int main() {
  uint64_t* d;
  hipMalloc(&d, 64);
  const char* names[3] = {"32 s_add", "32 v_add", "16 s_add + 16 v_add"};
  for (int kind = 0; kind < 3; kind++)
    for (int threads : {64, 256, 512, 1024}) {
      for (int rep = 0; rep < 2; rep++) {
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, d);
        if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, d);
        if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, d);
        hipDeviceSynchronize();
      }
      uint64_t h[2];
      hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      const double cyc = (double)h[0] / ROUNDS;  // cycles per round of 32 instructions, one wave's view
      const int waves = threads / 64;
      printf("%-20s %2d waves/CU (%d per SIMD): %.1f cycles per 32 instr per wave -> %.2f instr/cycle/CU\n", names[kind], waves, (waves + 3) / 4, cyc,
             32.0 * waves / cyc);
    }
  return 0;
}
