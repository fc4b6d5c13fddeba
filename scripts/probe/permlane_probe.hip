// Probe of gfx950 cross-lane primitives: prints the lane layout of v_permlane32_swap,
// v_permlane16_swap and DPP row_ror so the wave reduction can be written against facts.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  const unsigned l = threadIdx.x;
  unsigned a = l, b = 100 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[l] = r[0]; o[64 + l] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[128 + l] = q[0]; o[192 + l] = q[1];
  o[256 + l] = __builtin_amdgcn_update_dpp(0, (int)l, 0x128, 0xf, 0xf, true);
  o[320 + l] = __builtin_amdgcn_update_dpp(0, (int)l, 0x121, 0xf, 0xf, true);
}
int main() {
  unsigned* d; hipMalloc(&d, 384 * 4);
  k<<<1, 64>>>(d);
  unsigned h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[6] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1", "row_ror:8", "row_ror:1"};
  for (int s = 0; s < 6; ++s) { printf("%s:", names[s]); for (int i = 0; i < 64; ++i) printf(" %u", h[64 * s + i]); printf("\n"); }
  return 0;
}
