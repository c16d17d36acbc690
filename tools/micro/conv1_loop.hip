// Microbenchmark of the first-layer tile loop (conv1_fwd_u8_kernel): per 16-pixel tile 21 k-steps x 2 MFMAs, the
// B operand read from a byte image in LDS and converted on the fly.  Which part costs matrix-pipe time?
//   MODE 0: MFMAs only (B operand constant)
//   MODE 1: + 21 ds_read_u8 per tile (value used as raw bits, no conversion)
//   MODE 2: + v_cvt_f32_ubyte0 per element  (= the kernel's loop as the compiler schedules it)
//   MODE 3: float image in LDS: 21 ds_read_b32, no conversion
//   MODE 4: byte image, one ds_read_b32 feeds 4 consecutive k-steps through v_cvt_f32_ubyte0..3 (k = 16j+4kq+e order,
//           24 k-steps with the tap row padded to 32 columns)
//   MODE 5: MODE 2 with all 21 reads + conversions of the NEXT tile issued before the current tile's MFMAs
//   EPI  1: + bias/ReLU epilogue and the two 16-B global stores per lane per tile
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../curla_amd/csrc/common.h"

constexpr int NS = 21, KQ = 7, RSb = 704, C = 9;

template <int MODE, int EPI, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, const float* w, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  for (int i = tid; i < 14000; i += THREADS) lds[i] = (MODE == 3) ? (float)(i & 255) : __int_as_float(0x01020304 * (i & 31));
  constexpr int NK = MODE == 4 ? 24 : NS;
  float wr[NK][2];
#pragma unroll
  for (int s = 0; s < NK; ++s)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) wr[s][mt] = w[(s * 2 + mt) * 64 + lane];
  __syncthreads();
  const uint8_t* ldsb = reinterpret_cast<const uint8_t*>(lds);
  f32x4 tot = {0, 0, 0, 0};
  const f32x4 bias = {0.1f, 0.2f, 0.3f, 0.4f};
  float* o = out + ((size_t)blockIdx.x * THREADS + tid) * 8;
  float nb[NS];
  if (MODE == 5) {
    const uint8_t* base = ldsb + 18 * li + kq;
#pragma unroll
    for (int s = 0; s < NS; ++s) nb[s] = (float)base[(s / KQ) * RSb + 4 * (s % KQ)];
  }
  for (int tile = 0; tile < ntiles; ++tile) {
    const int ty = tile & 7, x = li + 16 * (tile & 1);
    f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        acc[0] = mfma16(wr[s][0], 1.5f + s, acc[0]);
        acc[1] = mfma16(wr[s][1], 1.5f + s, acc[1]);
      }
    } else if (MODE == 1 || MODE == 2) {
      const uint8_t* base = ldsb + 2 * ty * RSb + 2 * x * C + kq;
      float bv[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const uint8_t b = base[(s / KQ) * RSb + 4 * (s % KQ)];
        bv[s] = MODE == 2 ? (float)b : __int_as_float((int)b);
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        acc[0] = mfma16(wr[s][0], bv[s], acc[0]);
        acc[1] = mfma16(wr[s][1], bv[s], acc[1]);
      }
    } else if (MODE == 3) {
      const float* base = lds + 2 * ty * 688 + 2 * x * C + kq;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const float bv = base[(s / KQ) * 688 + 4 * (s % KQ)];
        acc[0] = mfma16(wr[s][0], bv, acc[0]);
        acc[1] = mfma16(wr[s][1], bv, acc[1]);
      }
    } else if (MODE == 4) {
      // per tap row: two aligned dwords per lane (bytes 16j+4kq .. +3 of the 32-byte padded run)
      const uint32_t* base = reinterpret_cast<const uint32_t*>(ldsb + 2 * ty * RSb + ((2 * x * C) & ~3)) + kq;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint32_t v = base[dy * (RSb / 4) + 4 * j];
          // (the compiler selects v_cvt_f32_ubyte0..3 for these byte extractions)
          const float b0 = (float)(v & 0xff), b1 = (float)((v >> 8) & 0xff);
          const float b2 = (float)((v >> 16) & 0xff), b3 = (float)(v >> 24);
          const int s = (dy * 2 + j) * 4;
          acc[0] = mfma16(wr[s][0], b0, acc[0]);
          acc[1] = mfma16(wr[s][1], b0, acc[1]);
          acc[0] = mfma16(wr[s + 1][0], b1, acc[0]);
          acc[1] = mfma16(wr[s + 1][1], b1, acc[1]);
          acc[0] = mfma16(wr[s + 2][0], b2, acc[0]);
          acc[1] = mfma16(wr[s + 2][1], b2, acc[1]);
          acc[0] = mfma16(wr[s + 3][0], b3, acc[0]);
          acc[1] = mfma16(wr[s + 3][1], b3, acc[1]);
        }
    } else {  // MODE 5
      float bv[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) bv[s] = nb[s];
      const int nty = (tile + 1) & 7, nx = li + 16 * ((tile + 1) & 1);
      const uint8_t* base = ldsb + 2 * nty * RSb + 2 * nx * C + kq;
#pragma unroll
      for (int s = 0; s < NS; ++s) nb[s] = (float)base[(s / KQ) * RSb + 4 * (s % KQ)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        acc[0] = mfma16(wr[s][0], bv[s], acc[0]);
        acc[1] = mfma16(wr[s][1], bv[s], acc[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (EPI) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 v = acc[mt] + bias;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(o + mt * 4));
      }
      o += (size_t)gridDim.x * THREADS * 8;
    } else {
      tot += acc[0] + acc[1];
    }
  }
  if (!EPI) out[(size_t)blockIdx.x * THREADS + tid] = tot[0] + tot[1] + tot[2] + tot[3];
}

template <int MODE, int EPI, int THREADS>
void run(const char* name, int blocks_per_cu) {
  float *out, *w;
  const int ntiles = EPI ? 64 : 2000, blocks = 256 * blocks_per_cu, lds = 56 * 1024;
  hipMalloc(&out, EPI ? (size_t)blocks * THREADS * 8 * 4 * ntiles : (size_t)blocks * THREADS * 4);
  hipMalloc(&w, 48 * 64 * 4);
  hipMemset(w, 0, 48 * 64 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, EPI, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, EPI, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, out, w, ntiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * (THREADS / 64) * ntiles * 42;  // useful MFMAs (MODE 4 issues 48)
    if (rep == 2)
      printf("%-58s %dx%d thr/CU %7.3f ms  %6.1f TF useful (%.1f %% of 157.3)\n", name, blocks_per_cu, THREADS, ms,
             mfma * 2048 / ms / 1e9, mfma * 2048 / ms / 1e9 / 157.3 * 100);
  }
  hipFree(out);
  hipFree(w);
}

int main() {
  for (int burn = 0; burn < 30; ++burn) run<0, 0, 512>(burn == 29 ? "MFMAs only" : "", 2);
  run<1, 0, 512>("+ ds_read_u8", 2);
  run<2, 0, 512>("+ cvt (kernel loop)", 2);
  run<3, 0, 512>("float image: ds_read_b32, no cvt", 2);
  run<4, 0, 512>("ds_read_b32 + cvt_ubyte0..3 (24 k-steps)", 2);
  run<5, 0, 512>("kernel loop, next tile's reads+cvt hoisted", 2);
  run<2, 1, 512>("kernel loop + epilogue stores", 2);
  run<3, 1, 512>("float image + epilogue stores", 2);
  run<5, 1, 512>("hoisted + epilogue stores", 2);
  run<2, 0, 256>("kernel loop, 256-thread WGs x2 (2 waves/SIMD)", 2);
  run<2, 0, 256>("kernel loop, 256-thread WGs x4 (4 waves/SIMD)", 4);
  run<3, 0, 256>("float image, 256-thread WGs x2", 2);
  run<5, 0, 256>("hoisted, 256-thread WGs x2", 2);
  run<0, 0, 256>("MFMAs only, 256 x2", 2);
  run<0, 0, 256>("MFMAs only, 256 x1 (1 wave/SIMD)", 1);
  run<2, 0, 256>("kernel loop, 256 x1 (1 wave/SIMD)", 1);
  run<5, 0, 256>("hoisted, 256 x1 (1 wave/SIMD)", 1);
  return 0;
}
