// Probe: one wave per SIMD (one 256-thread workgroup per CU, up to 512 registers per wave) computing BOTH
// 16-channel output halves from one transformed window: 32 MFMAs per 8 packed adds and 4 window reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../curla_amd/csrc/common.h"

__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(float* out, const float* w, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int li = lane & 15, kq = lane >> 4;
  for (int i = tid; i < 20000; i += 256) lds[i] = (float)(i & 7);
  float wu[2][3][4][8];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int s = 0; s < 8; ++s) wu[m][a][b][s] = w[(((m * 3 + a) * 4 + b) * 8 + s) * 64 + lane];
  __syncthreads();
  f32x4 tot = {0, 0, 0, 0};
  f32x2 wt = {0, 0};
  const int WT = 37;
  for (int tile = 0; tile < ntiles; ++tile) {
    const float* base = lds + ((tile & 7) * WT + 2 * li) * 36 + 4 * kq;
    f32x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[m][c] = f32x4{0, 0, 0, 0};
    f32x4 d[2][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) d[0][c] = *reinterpret_cast<const f32x4*>(base + c * 36);
#pragma unroll
    for (int h = 0; h < 6; ++h) {
      const int dy = h >> 1, q = h & 1;
      if (h < 5) {
        const int ndy = (h + 1) >> 1, nq = (h + 1) & 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) d[(h + 1) & 1][c] = *reinterpret_cast<const f32x4*>(base + (ndy * WT + c) * 36 + 16 * nq);
      }
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 d0 = d[h & 1][0], d1 = d[h & 1][1], d2 = d[h & 1][2], d3 = d[h & 1][3];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        f32x2 v0 = {d0[2 * p], d0[2 * p + 1]}, e1 = {d1[2 * p], d1[2 * p + 1]};
        f32x2 v2 = {d2[2 * p], d2[2 * p + 1]}, v3 = {d3[2 * p], d3[2 * p + 1]};
        winograd_bt_pk(v0, e1, v2, v3, wt);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int e = 2 * p + r;
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            acc[m][0] = mfma16(wu[m][dy][0][4 * q + e], v0[r], acc[m][0]);
            acc[m][1] = mfma16(wu[m][dy][1][4 * q + e], wt[r], acc[m][1]);
            acc[m][2] = mfma16(wu[m][dy][2][4 * q + e], v2[r], acc[m][2]);
            acc[m][3] = mfma16(wu[m][dy][3][4 * q + e], v3[r], acc[m][3]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) tot += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
  }
  out[blockIdx.x * 256 + tid] = tot[0] + tot[1] + tot[2] + tot[3];
}

int main() {
  float *out, *w;
  (void)hipMalloc(&out, 1024 * 256 * 4);
  (void)hipMalloc(&w, 192 * 64 * 4);
  (void)hipMemset(w, 0, 192 * 64 * 4);
  const int ntiles = 4000, blocks = 256, lds = 150 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, out, w, ntiles);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * 4 * ntiles * 192;
    if (rep == 2)
      printf("one wave per SIMD, both channel halves per wave: %7.2f ms  %5.1f %% of the MFMA peak, %5.1f cycles per MFMA per SIMD at 2.4 GHz\n",
             ms, mfma * 2048 / ms / 1e9 / 157.3 * 100, ms * 1e-3 * 2.4e9 / (mfma / 1024));
  }
  return 0;
}
