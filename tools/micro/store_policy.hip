// What the WRITE_SIZE counter says about the conv epilogue's stores (conv_rw_fwd_kernel writes 1.31x its
// output bytes).  The epilogue of the row-walk kernels stores an output pixel's 32 channels as TWO 64-byte half lines
// (one per 16-channel MFMA tile half), each by its own buffer_store_dwordx4 with the streaming policy (aux = 2: the data
// is not read again by the kernel).  This program writes the same [pixel][32 float] rows four ways and nothing else:
//   half_nt    64-byte half lines, streaming policy (what conv_rw.h does): lanes (li, kq) -> pixel 2 li, bytes 16 kq + 64 mt
//   half_wb    the same addresses with the default write-back policy
//   line_nt    whole 128-byte lines per store instruction (8 lanes x 16 bytes per pixel), streaming policy
//   line_wb    whole lines, write-back
// Run it under `rocprofv3 --pmc WRITE_SIZE --kernel-trace` (tools/pmc_store_policy.sh): the counter's excess over the
// bytes stored belongs to the (half line, streaming) combination, not to the kernel's arithmetic or its addressing.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/_build/store_policy tools/micro/store_policy.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(p, (short)0, (int)bytes, 0x00020000);
}

// rows: pixels of 128 bytes.  One wave covers 32 pixels per step (16 lanes li x 2 pixels), as the F(2,3) epilogue does.
template <int NT>
__global__ __launch_bounds__(256) void half_lines(float* out, unsigned npix) {
  const unsigned lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const __amdgpu_buffer_rsrc_t r = rsrc(out, npix * 128u);
  const u32x4 v = {lane, wave, 1u, 2u};
  for (unsigned p0 = wave * 32; p0 + 32 <= npix; p0 += nwaves * 32) {
    const unsigned oa = (p0 + 2 * li) * 128u + kq * 16u, ob = oa + 128u;
#pragma unroll
    for (unsigned mt = 0; mt < 2; ++mt) {
      __builtin_amdgcn_raw_buffer_store_b128(v, r, oa + mt * 64u, 0, NT ? 2 : 0);
      __builtin_amdgcn_raw_buffer_store_b128(v, r, ob + mt * 64u, 0, NT ? 2 : 0);
    }
  }
}

// the same bytes, but every store instruction writes whole lines: 8 consecutive lanes x 16 bytes = one pixel
template <int NT>
__global__ __launch_bounds__(256) void whole_lines(float* out, unsigned npix) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const __amdgpu_buffer_rsrc_t r = rsrc(out, npix * 128u);
  const u32x4 v = {lane, wave, 1u, 2u};
  for (unsigned p0 = wave * 32; p0 + 32 <= npix; p0 += nwaves * 32) {
#pragma unroll
    for (unsigned s = 0; s < 4; ++s)  // 4 instructions x 8 pixels
      __builtin_amdgcn_raw_buffer_store_b128(v, r, (p0 + 8 * s + (lane >> 3)) * 128u + (lane & 7) * 16u, 0, NT ? 2 : 0);
  }
}

int main() {
  const unsigned npix = 4u << 20;  // 4 Mi pixels x 128 bytes = 512 MiB (the c2 critic-phase stack writes 537 MB)
  float* out;
  if (hipMalloc(&out, (size_t)npix * 128) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  struct {
    const char* name;
    void (*k)(float*, unsigned);
  } ks[] = {{"half_nt", half_lines<1>}, {"half_wb", half_lines<0>}, {"line_nt", whole_lines<1>}, {"line_wb", whole_lines<0>}};
  for (auto& k : ks) {
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k.k, dim3(2048), dim3(256), 0, 0, out, npix);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("%-8s %8.1f us  %6.0f GB/s stored\n", k.name, ms * 1e3, (double)npix * 128 / ms / 1e6);
    }
  }
  return 0;
}
