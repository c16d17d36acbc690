// Microbenchmark of the conv_s1 tile loop (Winograd F(2,3) half-steps: 4 ds_read_b128, 8 packed adds, 16 MFMAs)
// with its parts switched off one at a time: which of them costs matrix-pipe time?
//   bit 1: LDS window reads   bit 2: packed input transform   (MFMAs always on)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../curla_amd/csrc/common.h"

// the same transform with plain (unpacked) VALU adds: 8 instructions per channel pair instead of 4 packed ones
// (x = d0 -> d0-d2, y = d2 -> d2-d1, z = d3 -> d1-d3, t -> d1+d2; two channels a/b)
__device__ __forceinline__ void winograd_bt_scalar(float& xa, float& xb, float& ya, float& yb, float& za, float& zb,
                                                   float& ta, float& tb, float d1a, float d1b) {
  asm("v_sub_f32 %0, %0, %2\n\t"
      "v_sub_f32 %1, %1, %3\n\t"
      "v_sub_f32 %4, %8, %4\n\t"
      "v_sub_f32 %5, %9, %5\n\t"
      "v_add_f32 %6, %8, %2\n\t"
      "v_add_f32 %7, %9, %3\n\t"
      "v_sub_f32 %2, %2, %8\n\t"
      "v_sub_f32 %3, %3, %9\n\t"
      "s_nop 1"
      : "+v"(xa), "+v"(xb), "+v"(ya), "+v"(yb), "+v"(za), "+v"(zb), "+v"(ta), "+v"(tb)
      : "v"(d1a), "v"(d1b));
}

template <int PARTS>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* w, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int li = lane & 15, kq = lane >> 4;
  for (int i = tid; i < 20000; i += 256) lds[i] = (float)((i * 2654435761u >> 20) & 255) * 0.01f;
  float wu[3][4][8];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int s = 0; s < 8; ++s) wu[a][b][s] = w[((a * 4 + b) * 8 + s) * 64 + lane];
  __syncthreads();
  f32x4 tot = {0, 0, 0, 0};
  f32x2 wt = {0, 0};
  const int WT = 37;
  if (PARTS & 8) {
    // the two waves that share a SIMD (one from each of the CU's two workgroups) sit in different wave slots: raise
    // the priority of the odd slot, so that the arbiter hands the matrix pipe to ONE of them whenever both want it
    // and the other's VALU / LDS work falls into that one's MFMA runs instead of coinciding with its own
    const unsigned hw = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);  // HW_REG_HW_ID, wave_id[3:0]
    if (hw & 1) __builtin_amdgcn_s_setprio(3);
  }
  if (PARTS & 16) {
    if (blockIdx.x & 1) __builtin_amdgcn_s_setprio(3);
  }
  for (int tile = 0; tile < ntiles; ++tile) {
    const float* base = lds + ((tile & 7) * WT + 2 * li) * 36 + 4 * kq;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x4 d[2][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) d[0][c] = (PARTS & 1) ? *reinterpret_cast<const f32x4*>(base + c * 36) : f32x4{1.f + c, 2, 3, 4};
#pragma unroll
    for (int h = 0; h < 6; ++h) {
      const int dy = h >> 1, q = h & 1;
      if (h < 5) {
        const int ndy = (h + 1) >> 1, nq = (h + 1) & 1;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          d[(h + 1) & 1][c] = (PARTS & 1) ? *reinterpret_cast<const f32x4*>(base + (ndy * WT + c) * 36 + 16 * nq)
                                          : f32x4{1.f + c + h, 2, 3, 4};
      }
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 d0 = d[h & 1][0], d1 = d[h & 1][1], d2 = d[h & 1][2], d3 = d[h & 1][3];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        f32x2 v0 = {d0[2 * p], d0[2 * p + 1]}, e1 = {d1[2 * p], d1[2 * p + 1]};
        f32x2 v2 = {d2[2 * p], d2[2 * p + 1]}, v3 = {d3[2 * p], d3[2 * p + 1]};
        if (PARTS & 4) {
          float xa = v0[0], xb = v0[1], ya = v2[0], yb = v2[1], za = v3[0], zb = v3[1], ta = wt[0], tb = wt[1];
          winograd_bt_scalar(xa, xb, ya, yb, za, zb, ta, tb, e1[0], e1[1]);
          v0 = f32x2{xa, xb}, v2 = f32x2{ya, yb}, v3 = f32x2{za, zb}, wt = f32x2{ta, tb};
        } else if (PARTS & 2) {
          winograd_bt_pk(v0, e1, v2, v3, wt);
        } else {
          wt = e1;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int e = 2 * p + r;
          acc[0] = mfma16(wu[dy][0][4 * q + e], v0[r], acc[0]);
          acc[1] = mfma16(wu[dy][1][4 * q + e], wt[r], acc[1]);
          acc[2] = mfma16(wu[dy][2][4 * q + e], v2[r], acc[2]);
          acc[3] = mfma16(wu[dy][3][4 * q + e], v3[r], acc[3]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    tot += acc[0] + acc[1] + acc[2] + acc[3];
  }
  out[blockIdx.x * 256 + tid] = tot[0] + tot[1] + tot[2] + tot[3];
}

template <int PARTS>
void run(const char* name) {
  float *out, *w;
  hipMalloc(&out, 1024 * 256 * 4);
  hipMalloc(&w, 96 * 64 * 4);
  {  // random weights: nothing about the timing may depend on all-zero operands
    std::vector<float> hw(96 * 64);
    unsigned st = 12345u;
    for (auto& v : hw) st = st * 1664525u + 1013904223u, v = ((st >> 8) & 0xffff) / 65536.0f - 0.5f;
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  }
  const int ntiles = 4000, blocks = 512, lds = 80 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<PARTS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<PARTS>, dim3(blocks), dim3(256), lds, 0, out, w, ntiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * 4 * ntiles * 96;  // MFMAs
    if (rep == 2)
      printf("%-34s %7.2f ms  %6.1f TF of MFMA work (%.1f %% of 157.3), %5.1f cycles per MFMA per SIMD at 2.4 GHz\n", name, ms,
             mfma * 2048 / ms / 1e9, mfma * 2048 / ms / 1e9 / 157.3 * 100, ms * 1e-3 * 2.4e9 / (mfma / 1024));
  }
}

int main() {
  run<0>("MFMAs only");
  run<1>("+ LDS window reads");
  run<2>("+ packed input transform");
  run<3>("+ both (the kernel's tile loop)");
  run<3 + 8>("tile loop + s_setprio on the odd wave slot");
  run<3 + 16>("tile loop + s_setprio on odd blockIdx");
  run<5>("reads + UNPACKED transform (8 v_add/v_sub)");
  run<4>("UNPACKED transform only");
  return 0;
}
