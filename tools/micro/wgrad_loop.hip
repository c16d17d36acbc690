// Microbenchmark of the wgrad_s1 k-step loop (per 4 pixel pairs: 2 + 12 LDS reads, 12 packed adds, the pair
// walk, 24 MFMAs) with its parts switched off one at a time.
//   bit 1: LDS reads   bit 2: packed Winograd transform   bit 4: pair-walk arithmetic (addresses, predicates)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../curla_amd/csrc/common.h"

template <int PARTS>
__global__ __launch_bounds__(256, 2) void k(float* out, int nunits, int Wi, int Wo) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int mt = wave & 1, uslot = wave >> 1;
  for (int i = tid; i < 20000; i += 256) lds[i] = (float)(i & 7);
  __syncthreads();
  f32x4 acc[3][4][2];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[a][b][c] = f32x4{0, 0, 0, 0};
  f32x2 wt = {0, 0};
  float bsum = 0.f;
  const int PW = (Wo + 1) >> 1, npairs = 6 * PW;
  const float* ldsg = lds + 8 * Wi * 36;
  int fy = ((uslot & 1) + 2 * kq) / PW, fj = ((uslot & 1) + 2 * kq) - fy * PW;
  auto fetch = [&](int u, float (&gv)[2], f32x2 (&dv)[3][4]) {
    int ty = 0, x0 = 0;
    bool pv = true;
    if (PARTS & 4) {
      const int q = ((u >> 1) * 8 + (u & 1) + 2 * kq) % npairs;
      pv = q < npairs;
      ty = pv ? fy : 0, x0 = pv ? 2 * fj : 0;
    }
    const float* gp = ldsg + (ty * Wo + x0) * 36 + mt * 16 + li;
    const float* ip = lds + (ty * Wi + x0) * 36 + li;
    if (PARTS & 1) {
      gv[0] = pv ? gp[0] : 0.f;
      gv[1] = (pv && x0 + 1 < Wo) ? gp[36] : 0.f;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float* qd = ip + (dy * Wi + c) * 36;
          dv[dy][c] = f32x2{qd[0], qd[16]};
        }
    } else {
      gv[0] = 1.f + u, gv[1] = 2.f;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int c = 0; c < 4; ++c) dv[dy][c] = f32x2{1.f + c + u, 2.f + dy};
    }
    if (PARTS & 4) {
      fj += 8;
      const bool wrap = fj >= PW;
      fj = wrap ? fj - PW : fj;
      fy = wrap ? fy + 1 : fy;
      fy = fy >= 6 ? 0 : fy;
    }
  };
  auto mma = [&](const float (&gv)[2], f32x2 (&dv)[3][4]) {
    bsum += gv[0] + gv[1];
    const float g0 = gv[0], g1 = gv[0] + gv[1], g2 = gv[0] - gv[1], g3 = -gv[1];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      if (PARTS & 2) winograd_bt_pk(dv[dy][0], dv[dy][1], dv[dy][2], dv[dy][3], wt); else wt = dv[dy][1];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        acc[dy][0][ct] = mfma16(g0, dv[dy][0][ct], acc[dy][0][ct]);
        acc[dy][1][ct] = mfma16(g1, wt[ct], acc[dy][1][ct]);
        acc[dy][2][ct] = mfma16(g2, dv[dy][2][ct], acc[dy][2][ct]);
        acc[dy][3][ct] = mfma16(g3, dv[dy][3][ct], acc[dy][3][ct]);
      }
    }
  };
  float gA[2], gB[2];
  f32x2 dA[3][4], dB[3][4];
  fetch(uslot, gA, dA);
  for (int u = uslot; u < nunits; u += 4) {
    fetch(u + 2, gB, dB);
    __builtin_amdgcn_sched_barrier(0);
    mma(gA, dA);
    __builtin_amdgcn_sched_barrier(0);
    fetch(u + 4, gA, dA);
    __builtin_amdgcn_sched_barrier(0);
    mma(gB, dB);
    __builtin_amdgcn_sched_barrier(0);
  }
  f32x4 tot = {bsum, 0, 0, 0};
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) tot += acc[a][b][c];
  out[blockIdx.x * 256 + tid] = tot[0] + tot[1] + tot[2] + tot[3];
}

template <int PARTS>
void run(const char* name) {
  float* out;
  (void)hipMalloc(&out, 1024 * 256 * 4);
  const int nunits = 40000, blocks = 512, lds = 80 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<PARTS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<PARTS>, dim3(blocks), dim3(256), lds, 0, out, nunits, 37, 35);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * 4 * (nunits / 2) * 24;
    if (rep == 2)
      printf("%-44s %7.2f ms  %5.1f %% of the MFMA peak, %5.1f cycles per MFMA per SIMD at 2.4 GHz\n", name, ms,
             mfma * 2048 / ms / 1e9 / 157.3 * 100, ms * 1e-3 * 2.4e9 / (mfma / 1024));
  }
}

int main() {
  run<0>("MFMAs only");
  run<1>("+ LDS reads (2 b32 + 12 read2_b32 per 24 MFMAs)");
  run<2>("+ packed transform (12 adds per 24 MFMAs)");
  run<4>("+ pair walk");
  run<3>("+ LDS reads + transform");
  run<7>("all (the kernel's k-step loop)");
  return 0;
}
