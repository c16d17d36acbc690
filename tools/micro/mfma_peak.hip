// Microbenchmark: issue rate of v_mfma_f32_16x16x4_f32 on MI355X (chains per wave, waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
void run(int blocks, int threads, const char* name) {
  float* out;
  hipMalloc(&out, (size_t)blocks * threads * 4);
  int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)blocks * (threads / 64) * iters * 16.0 * CHAINS * 2048.0;
  printf("%-40s %8.3f ms  %7.1f TF\n", name, ms, flop / ms / 1e9);
  hipFree(out);
}
int main() {
  run<1>(256, 256, "1 chain, 1 wave/SIMD");
  run<2>(256, 256, "2 chains, 1 wave/SIMD");
  run<4>(256, 256, "4 chains, 1 wave/SIMD");
  run<1>(512, 256, "1 chain, 2 waves/SIMD");
  run<2>(512, 256, "2 chains, 2 waves/SIMD");
  run<2>(1024, 256, "2 chains, 4 waves/SIMD");
  return 0;
}
