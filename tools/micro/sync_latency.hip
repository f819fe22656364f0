// How long does the host wait for a finished stream?  hipStreamSynchronize against polling hipStreamQuery, after a
// short kernel + a 16-byte device-to-host copy (what every zes_* device entry point ends with).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/sync_latency.hip -o gpurun_out/sync_latency && gpurun_out/sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long* out, unsigned cycles) {
  const unsigned long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (threadIdx.x == 0) out[0] = t0;
}
int main() {
  unsigned long long* d;
  void* h;
  hipStream_t s;
  hipMalloc(&d, 64);
  hipHostMalloc(&h, 64, hipHostMallocDefault);
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (unsigned cycles : {1000u, 100000u, 1000000u}) {
    for (int mode = 0; mode < 2; mode++) {
      double tot = 0;
      const int N = 300;
      for (int i = 0; i < N + 10; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, cycles);
        hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, s);
        if (mode == 0) hipStreamSynchronize(s);
        else
          while (hipStreamQuery(s) == hipErrorNotReady) {}
        const auto t1 = std::chrono::steady_clock::now();
        if (i >= 10) tot += std::chrono::duration<double, std::micro>(t1 - t0).count();
      }
      printf("kernel ~%u cycles, %s: %.1f us per launch+copy+wait\n", cycles, mode ? "hipStreamQuery poll" : "hipStreamSynchronize", tot / N);
    }
  }
  return 0;
}
