// checks the inline-asm packed add/sub helpers of common.h on the device
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../curla_amd/csrc/common.h"
__global__ void k(const float* a, const float* b, float* c) {
  int i = threadIdx.x;
  f32x2 x = {a[2 * i], a[2 * i + 1]}, y = {b[2 * i], b[2 * i + 1]};
  f32x2 s = pk_sub(x, y), t = pk_add(x, y);
  c[4 * i] = s[0], c[4 * i + 1] = s[1], c[4 * i + 2] = t[0], c[4 * i + 3] = t[1];
}
int main() {
  float ha[128], hb[128], hc[256];
  for (int i = 0; i < 128; ++i) ha[i] = i * 1.5f, hb[i] = 100.f - i;
  float *a, *b, *c;
  hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&c, 1024);
  hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, c);
  hipMemcpy(hc, c, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i)
    for (int r = 0; r < 2; ++r) {
      if (hc[4 * i + r] != ha[2 * i + r] - hb[2 * i + r]) ++bad;
      if (hc[4 * i + 2 + r] != ha[2 * i + r] + hb[2 * i + r]) ++bad;
    }
  printf("bad=%d  c[0..3]=%g %g %g %g (want %g %g %g %g)\n", bad, hc[0], hc[1], hc[2], hc[3], ha[0] - hb[0], ha[1] - hb[1], ha[0] + hb[0], ha[1] + hb[1]);
  return bad != 0;
}
