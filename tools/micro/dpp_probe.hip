// dpp_probe.hip — which lane does each cross-lane form used by zes_index.hip read from?  (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/dpp_probe.hip -o tools/micro/dpp_probe && tools/micro/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int CTRL, int BANK>
__device__ uint32_t dpp(uint32_t old, uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)x, CTRL, 0xf, BANK, false); }
__global__ void probe(uint32_t* out) {
  const uint32_t lane = threadIdx.x;
  out[0 * 64 + lane] = dpp<0xB1, 0xf>(999, lane);
  out[1 * 64 + lane] = dpp<0x4E, 0xf>(999, lane);
  out[2 * 64 + lane] = dpp<0x124, 0xf>(999, lane);
  out[3 * 64 + lane] = dpp<0x12C, 0xf>(999, lane);
  out[4 * 64 + lane] = dpp<0x128, 0xf>(999, lane);
  out[5 * 64 + lane] = (uint32_t)__builtin_amdgcn_ds_swizzle((int)lane, 0x401F);
  out[6 * 64 + lane] = dpp<0x138, 0xf>(999, lane);
  out[7 * 64 + lane] = dpp<0x124, 0xA>(999, lane);
  out[8 * 64 + lane] = dpp<0x111, 0xf>(999, lane);
}
int main() {
  uint32_t* d;
  hipMalloc(&d, 9 * 64 * 4);
  probe<<<1, 64>>>(d);
  uint32_t h[9 * 64];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* nm[9] = {"quad[1,0,3,2]", "quad[2,3,0,1]", "row_ror:4", "row_ror:12", "row_ror:8", "swizzle xor16", "wave_shr:1", "row_ror:4 bank A", "row_shr:1"};
  for (int r = 0; r < 9; r++) {
    printf("%-18s", nm[r]);
    for (int l = 0; l < 64; l++) printf(" %u", h[r * 64 + l]);
    printf("\n");
  }
  return 0;
}
