// Probe: wave_reduce4 of olsr_device.h on known inputs, step by step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../online_lang_splatting_amd/csrc/olsr_device.h"
using namespace olsr;
__global__ void k(float* o, const float* in) {
  const int l = threadIdx.x;
  const float a = in[l], b = in[64 + l], c = in[128 + l], d = in[192 + l];
  const float s0 = swap32_add(a, b);
  const float s1 = swap32_add(c, d);
  const float t0 = swap16_add(s0, s1);
  o[l] = s0; o[64 + l] = s1; o[128 + l] = t0;
  o[192 + l] = wave_reduce4(a, b, c, d);
}
int main() {
  float *d, *in; (void)hipMalloc(&d, 256 * 4); (void)hipMalloc(&in, 256 * 4);
  float hin[256];
  for (int i = 0; i < 64; ++i) { hin[i] = (float)i; hin[64 + i] = 1000.f + i; hin[128 + i] = 0.5f * i; hin[192 + i] = 2.f; }
  (void)hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, in);
  float h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* n[4] = {"s0", "s1", "t0", "r4"};
  for (int s = 0; s < 4; ++s) { printf("%s:", n[s]); for (int i = 0; i < 64; ++i) printf(" %g", h[64 * s + i]); printf("\n"); }
  printf("expect r4 rows: %g %g %g %g\n", 2016.f, 0.5f * 2016.f, 64000.f + 2016.f, 128.f);
  return 0;
}
