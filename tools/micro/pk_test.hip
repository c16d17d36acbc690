// checks common.h's packed Winograd input transform (inline-asm block) on the device against scalar arithmetic
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../curla_amd/csrc/common.h"
__global__ void k(const float* in, float* out) {
  const int i = threadIdx.x;
  f32x2 d0 = {in[8 * i], in[8 * i + 1]}, d1 = {in[8 * i + 2], in[8 * i + 3]};
  f32x2 d2 = {in[8 * i + 4], in[8 * i + 5]}, d3 = {in[8 * i + 6], in[8 * i + 7]}, t = {0, 0};
  winograd_bt_pk(d0, d1, d2, d3, t);  // d0 <- d0-d2, t <- d1+d2, d2 <- d2-d1, d3 <- d1-d3
  out[8 * i] = d0[0], out[8 * i + 1] = d0[1], out[8 * i + 2] = t[0], out[8 * i + 3] = t[1];
  out[8 * i + 4] = d2[0], out[8 * i + 5] = d2[1], out[8 * i + 6] = d3[0], out[8 * i + 7] = d3[1];
}
int main() {
  float h[512], o[512];
  for (int i = 0; i < 512; ++i) h[i] = (float)((i * 37) % 101) * 0.25f - 7.f;
  float *a, *b;
  (void)hipMalloc(&a, sizeof h);
  (void)hipMalloc(&b, sizeof o);
  (void)hipMemcpy(a, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b);
  (void)hipMemcpy(o, b, sizeof o, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i)
    for (int r = 0; r < 2; ++r) {
      const float d0 = h[8 * i + r], d1 = h[8 * i + 2 + r], d2 = h[8 * i + 4 + r], d3 = h[8 * i + 6 + r];
      bad += o[8 * i + r] != d0 - d2;
      bad += o[8 * i + 2 + r] != d1 + d2;
      bad += o[8 * i + 4 + r] != d2 - d1;
      bad += o[8 * i + 6 + r] != d1 - d3;
    }
  printf("winograd_bt_pk: %d mismatches of 512\n", bad);
  return bad != 0;
}
