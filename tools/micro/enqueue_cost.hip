// Microbenchmark: host-side cost of enqueueing small stream operations (what a call with many tiny memsets / copies pays
// while the GPU waits for work).  build: hipcc -O3 --offload-arch=gfx950 enqueue_cost.hip -o enqueue_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_nop(unsigned* p) { if (threadIdx.x == 999) p[0] = 1; }
int main() {
  unsigned* d; void* h;
  hipMalloc(&d, 1 << 20); hipHostMalloc(&h, 1 << 20, hipHostMallocDefault);
  hipStream_t s; hipStreamCreate(&s);
  auto run = [&](const char* name, auto f) {
    for (int i = 0; i < 50; i++) f();
    hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 1000; i++) f();
    const auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s);
    const auto t2 = std::chrono::steady_clock::now();
    printf("%-38s enqueue %6.2f us each, drained %6.2f us each\n", name, std::chrono::duration<double, std::micro>(t1 - t0).count() / 1000,
           std::chrono::duration<double, std::micro>(t2 - t0).count() / 1000);
  };
  run("kernel launch (1 x 64)", [&] { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, d); });
  run("hipMemsetAsync 4 B", [&] { hipMemsetAsync(d, 0, 4, s); });
  run("hipMemsetAsync 8 KiB", [&] { hipMemsetAsync(d, 0xFF, 8192, s); });
  run("hipMemcpyAsync H2D 64 B (pinned)", [&] { hipMemcpyAsync(d, h, 64, hipMemcpyHostToDevice, s); });
  run("hipMemcpyAsync D2H 64 B (pinned)", [&] { hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, s); });
  // a dependent pair with a synchronisation in between: what one host round trip costs
  {
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 200; i++) {
      hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, d);
      hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, s);
      hipStreamSynchronize(s);
    }
    const auto t1 = std::chrono::steady_clock::now();
    printf("kernel + D2H 64 B + synchronize:        %6.2f us per round trip\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 200);
  }
  return 0;
}
