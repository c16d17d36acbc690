// Microbenchmark: what limits the conv tile loop (72 weight registers x MFMA 16x16x4 f32)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// VAR 1: 72 distinct A regs, B constant regs.  VAR 2: B from LDS reads (b128, pipelined like the kernel).
template <int VAR, int NCHAIN>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* w, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int li = lane & 15, kq = lane >> 4;
  for (int i = tid; i < 20480; i += 256) lds[i] = (float)(i & 7);
  float wr[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int s = 0; s < 8; ++s) wr[t][s] = w[(t * 8 + s) * 64 + lane];
  __syncthreads();
  f32x4 tot = {0, 0, 0, 0};
  for (int tile = 0; tile < ntiles; ++tile) {
    const float* base = lds + ((tile & 7) * 37 + li) * 40 + 4 * kq;
    f32x4 acc[NCHAIN];
#pragma unroll
    for (int c = 0; c < NCHAIN; ++c) acc[c] = f32x4{0, 0, 0, 0};
    f32x4 nb0, nb1;
    if (VAR == 2) {
      nb0 = *reinterpret_cast<const f32x4*>(base);
      nb1 = *reinterpret_cast<const f32x4*>(base + 16);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    } else {
      nb0 = f32x4{1, 2, 3, 4};
      nb1 = f32x4{5, 6, 7, 8};
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const f32x4 b0 = nb0, b1 = nb1;
      if (VAR == 2 && t < 8) {
        const int dy = (t + 1) / 3, dx = (t + 1) - 3 * dy;
        const float* ptr = base + (dy * 37 + dx) * 40;
        nb0 = *reinterpret_cast<const f32x4*>(ptr);
        nb1 = *reinterpret_cast<const f32x4*>(ptr + 16);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e % NCHAIN] = mfma16(wr[t][e], b0[e], acc[e % NCHAIN]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e % NCHAIN] = mfma16(wr[t][4 + e], b1[e], acc[e % NCHAIN]);
      if (VAR == 2) {
        if (t < 8) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      }
    }
#pragma unroll
    for (int c = 0; c < NCHAIN; ++c) tot += acc[c];
  }
  out[blockIdx.x * 256 + tid] = tot[0] + tot[1] + tot[2] + tot[3];
}

template <int VAR, int NCHAIN>
void run(int blocks, const char* name) {
  float *out, *w;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipMalloc(&w, 72 * 64 * 4);
  hipMemset(w, 0, 72 * 64 * 4);
  const int ntiles = 400;
  const size_t lds = 80 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<VAR, NCHAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<VAR, NCHAIN>), dim3(blocks), dim3(256), lds, 0, out, w, 4);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<VAR, NCHAIN>), dim3(blocks), dim3(256), lds, 0, out, w, ntiles);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)blocks * 4 * ntiles * 72.0 * 2048.0;
  printf("%-52s %8.3f ms  %7.1f TF\n", name, ms, flop / ms / 1e9);
}
int main() {
  run<1, 2>(512, "regs only, 2 chains, 2 WG/CU");
  run<1, 4>(512, "regs only, 4 chains, 2 WG/CU");
  run<1, 2>(256, "regs only, 2 chains, 1 WG/CU");
  run<2, 2>(512, "LDS b128 pipelined, 2 chains, 2 WG/CU");
  run<2, 4>(512, "LDS b128 pipelined, 4 chains, 2 WG/CU");
  run<2, 2>(256, "LDS b128 pipelined, 2 chains, 1 WG/CU");
  return 0;
}
