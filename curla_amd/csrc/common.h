// Shared device helpers for the curla_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/curla_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// 1-D Winograd input transform B^T d = (d0-d2, d1+d2, d2-d1, d1-d3) on two channels at once with packed fp32
// adds (4 VALU issues instead of 8; the compiler scalarises vector arithmetic whose components feed separate
// MFMAs).  In place: v0 -> d0, v2 -> d2, v3 -> d3, v1 -> t.  The block is inline asm, which the compiler's hazard
// recogniser cannot look into, so the two MFMA hazards a VALU write can hit are handled here by construction:
//   * VALU write -> MFMA read within 2 wait states: the trailing s_nop 1;
//   * MFMA SrcC read -> VALU write of that VGPR within 7 wait states: every register written here is either an
//     LDS-read destination (d0, d2, d3: the read's latency separates it from any earlier MFMA) or `t`, which the
//     caller keeps alive for the whole kernel so that it can never be a just-released accumulator.
// tools/check_asm_hazards.py scans the generated ISA for both (tests/test_capi.py runs it).
__device__ __forceinline__ void winograd_bt_pk(f32x2& d0, const f32x2& d1, f32x2& d2, f32x2& d3, f32x2& t) {
  asm("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %2, %4, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %3, %4, %1\n\t"
      "v_pk_add_f32 %1, %1, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "s_nop 1"
      : "+v"(d0), "+v"(d2), "+v"(d3), "+v"(t)
      : "v"(d1));
}

// Cache policy (the `aux` operand) of the conv epilogues' activation stores.  A pixel's 32 channels leave a wave as TWO
// 64-byte half lines (one per 16-channel MFMA tile half) from two store instructions.  With the streaming policy (2) such
// half-line stores cost 1.48 x their bytes at the memory side and run at 2.9 TB/s when nothing else is going on; with the
// default write-back policy (0) the L2 merges the halves: 1.00 x, 5.4 TB/s (tools/micro/store_policy.hip,
// profiles/r04_checks/r04_store_policy.txt) -- and a layer's output is what the next layer of the same launch, or the
// next launch, reads first.
#ifndef CURLA_ACT_STORE_POLICY
#define CURLA_ACT_STORE_POLICY 0
#endif

// (the flat-address form of the same choice, for the banded first-layer kernels)
__device__ __forceinline__ void act_store(f32x4* p, f32x4 v) {
  if (CURLA_ACT_STORE_POLICY)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

static inline int curla_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? CURLA_OK : CURLA_ERR_LAUNCH;
}

// compute units of the CURRENT device (cached per device id: a process may drive several GPUs)
static inline int curla_cu_count() {
  static int cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cached[dev];
  if (n == 0) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
    cached[dev] = n;
  }
  return n;
}

// Make sure a kernel's dynamic-LDS limit on the current device is at least `bytes`.  hipFuncSetAttribute is per-function,
// per-device process state (not stream-ordered): the largest size asked for so far is remembered per (kernel, device)
// and the limit is only ever RAISED -- a process that drives several GPUs configures each of them, two host threads
// cannot race each other's limit, and a launch that needs more than an earlier one of the same kernel gets it.
static inline int curla_set_dyn_lds(const void* fn, size_t bytes) {
  static std::mutex mu;
  static const void* done_fn[256];
  static int done_dev[256];
  static size_t done_bytes[256];
  static int ndone = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return CURLA_ERR_LAUNCH;
  std::lock_guard<std::mutex> lock(mu);
  int slot = -1;
  for (int i = 0; i < ndone; ++i)
    if (done_fn[i] == fn && done_dev[i] == dev) {
      if (bytes <= done_bytes[i]) return CURLA_OK;
      slot = i;
      break;
    }
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return CURLA_ERR_LAUNCH;
  if (slot < 0 && ndone < 256) slot = ndone++;
  if (slot >= 0) done_fn[slot] = fn, done_dev[slot] = dev, done_bytes[slot] = bytes;
  return CURLA_OK;
}

#define CURLA_REQUIRE(cond) \
  do {                      \
    if (!(cond)) return CURLA_ERR_ARG; \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
