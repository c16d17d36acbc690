// Shared device helpers for the curla_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/curla_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: D(16x16) += A(16x4) * B(4x16), exact f32 (a k-ordered
// fmaf chain).  Lane l holds A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
// D element (row = 4*(l>>4) + r, col = l&15) is register r of lane l.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

static inline int curla_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? CURLA_OK : CURLA_ERR_LAUNCH;
}

static inline int curla_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

#define CURLA_REQUIRE(cond) \
  do {                      \
    if (!(cond)) return CURLA_ERR_ARG; \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
