// 3x3 convolution kernels of the CNNEncoder hot path, gfx950.
//
// Reference semantics: encoder.py:54-63 (Conv2d k=3, first layer stride 2, the
// rest stride 1, no padding) + encoder.py:77-90 (obs/255, relu(conv)).  The
// backward kernels are the autograd of those lines.
//
// Layout: activations are NHWC fp32 in HBM ([B][H][W][32]); conv weights stay
// in the reference OIHW layout (they are nn.Parameters shared with Adam) and
// are re-gathered into MFMA operand images at kernel start.  The stride-1 forward and data gradient run on the bf16
// matrix cores with fp32 operands as exact sums of three bf16 parts (conv_rwb.h, round 5: six exact bf16 products per
// fp32 product, fp32 accumulation); every other inner product runs on the exact-f32 matrix pipe
// (v_mfma_f32_16x16x4_f32), which has the same 157 TFLOP/s roof as the fp32 vector pipe but needs one operand VGPR
// per lane instead of 2 per FMA.  The stride-1 layers (forward, data gradient, weight gradient) are row-walk
// kernels (conv_rwb.h; conv_rw.h / conv_rw43.h: the f32-input forms, selectable; conv_rw_wgrad2.h / conv_rw_wgrad.h): 1-D
// Winograd F(2,3) along x (the weight gradient: F(3,2) in both directions), a wave walks down a strip of pixel-pair
// columns with the transformed filter streamed from LDS; the first
// layer has a banded form (crop staged in LDS) and row-walk forms (conv1_rw.h, conv1_u8_rw.h).  What bounds the
// f32-input loops is VALU issue time (a VALU instruction and an f32 MFMA cannot issue in the same cycle), so
// everything in them is counted in instructions.  (Rounds 1-3 also
// carried a banded LDS-tiled form of the stride-1 kernels; it was removed in round 4 once the row walk covered
// every shape -- DESIGN.md section 3.)
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.h"
#include "options.h"

namespace {

constexpr int kLdsPix = 36;   // floats per pixel in LDS (32 channels + 4 pad: conflict-free b128 reads)

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

// 128 bytes of zeros in HBM: where a lane with nothing to multiply points its operand load (a select on the loaded
// VALUE would make the wave wait for the load right behind it instead of one or more k-steps later)
__device__ float g_zero_px[32];

// Timing-only ablations for tools/kbench.py (results are wrong when set); compiled out of the product library.
#ifdef CURLA_ABLATE
int g_ablate = 0;
#define ABL(bit) (a.dbg & (bit))
#define ABL_HOST g_ablate
#else
#define ABL(bit) 0
#define ABL_HOST 0
#endif

#include "conv_rw.h"

// Row-walk kernels (conv_rw.h).  Forward: 512-thread workgroups, ONE per CU, persistent over the samples they own;
// 96 KB of LDS hold the transformed filters of both problems of a layer, so the eight waves draw their steps from one
// pool (minibatch one and two together) and nothing but the layer boundary synchronises them.
__global__ __launch_bounds__(512, 2) void conv_rw_fwd_kernel(rw::Args A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int l = 0; l < A.nlayers; ++l) {
    rw::build_filter<MODE_FWD, 512>(lds, A.p[l][0].w, A.p[l][1].B > 0 ? A.p[l][1].w : nullptr, threadIdx.x);
    __syncthreads();
    rw::run_layer<MODE_FWD, 8>(A.g[l], A.p[l][0], A.p[l][1], lds, blockIdx.x, gridDim.x);
    if (l + 1 < A.nlayers) {
      // this workgroup's outputs of layer l are (only) its own inputs of layer l + 1; the barrier also keeps the
      // filter in LDS until every wave has finished reading it
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  }
}

#include "conv_rw43.h"

// The forward with Winograd F(4,3) along x (conv_rw43.h): 0.75 x the MFMAs; 256-thread workgroups, ONE wave per SIMD
// (the 144 accumulator + 96 window registers of a wave do not fit twice), 144 KB of LDS for the two filters.
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_rw43_fwd_kernel(rw::Args A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int l = 0; l < A.nlayers; ++l) {
    rw43::build_filter<MODE_FWD, 256>(lds, A.p[l][0].w, A.p[l][1].B > 0 ? A.p[l][1].w : nullptr, threadIdx.x);
    __syncthreads();
    rw43::run_layer<MODE_FWD, 4>(A.g[l], A.p[l][0], A.p[l][1], lds, blockIdx.x, gridDim.x);
    if (l + 1 < A.nlayers) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  }
}

// ... and the data gradient in the same form (one wave per SIMD: it cannot share a launch with the 256-register weight
// gradient, so wide layers run the two as two launches -- at their size a launch boundary is noise)
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_rw43_dgrad_kernel(rw::Args A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  rw43::build_filter<MODE_DGRAD, 256>(lds, A.p[0][0].w, nullptr, threadIdx.x);
  __syncthreads();
  rw43::run_layer<MODE_DGRAD, 4>(A.g[0], A.p[0][0], A.p[0][1], lds, blockIdx.x, gridDim.x);
}

#include "conv_rwb.h"

// The same layers on the bf16 matrix cores with fp32 operands split into three bf16 parts (conv_rwb.h): 512-thread
// workgroups, one per CU, 2 x rwb::kWBytes = 144 KB of LDS for the split filters of the two problems of a layer (72 KB
// each: 16 KB of headroom under the CU's 160 KB, asserted beside kMaxLds below).
#ifdef RWB_CLOCK
// diagnostic build (tools/clock_reconcile.sh: -DRWB_CLOCK): per workgroup, summed over launches, wave 0's shader cycles
// (s_memtime) and 100 MHz ticks (s_memrealtime) from its first instruction to its last, and the launch count.  Two
// stamps per workgroup and launch: nothing inside the loops (the per-step stamps of RWB_STAMP cost ~11 % of a wave).
__device__ unsigned long long g_rwb_clock[3 * 1024];
#endif

template <int NW>
__device__ __forceinline__ void rwb_fwd_body(const rw::Args& A) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds_h[];
#if defined(RWB_STAMP) || defined(RWB_CLOCK)
  const unsigned long long kt0 = __builtin_readcyclecounter(), kr0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int l = 0; l < A.nlayers; ++l) {
#ifdef RWB_STAMP
    const unsigned long long k0 = __builtin_readcyclecounter();
#endif
    rwb::build_filter<MODE_FWD, 64 * NW>(lds_h, A.p[l][0].w, A.p[l][1].B > 0 ? A.p[l][1].w : nullptr, threadIdx.x);
    __syncthreads();
#ifdef RWB_STAMP
    const unsigned long long k1 = __builtin_readcyclecounter();
#endif
    rwb::run_layer<MODE_FWD, NW>(A.g[l], A.p[l][0], A.p[l][1], lds_h, blockIdx.x, gridDim.x);
#ifdef RWB_STAMP
    const unsigned long long k2 = __builtin_readcyclecounter();
#endif
    if (l + 1 < A.nlayers) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
#ifdef RWB_STAMP
    const unsigned long long k3 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 7) {
      rwb::g_rwb_stamp[4] += k1 - k0, rwb::g_rwb_stamp[5] += k2 - k1, rwb::g_rwb_stamp[6] += k3 - k2, rwb::g_rwb_stamp[7] += 1;
    }
#endif
  }
#ifdef RWB_STAMP
  if (threadIdx.x == 0 && blockIdx.x == 7) {
    rwb::g_rwb_stamp[15] += __builtin_readcyclecounter() - kt0;
    rwb::g_rwb_stamp[14] += __builtin_amdgcn_s_memrealtime() - kr0;
  }
#endif
#ifdef RWB_CLOCK
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    g_rwb_clock[3 * blockIdx.x + 0] += __builtin_readcyclecounter() - kt0;
    g_rwb_clock[3 * blockIdx.x + 1] += __builtin_amdgcn_s_memrealtime() - kr0;
    g_rwb_clock[3 * blockIdx.x + 2] += 1;
  }
#endif
}

__global__ __launch_bounds__(512, 1) void conv_rwb_fwd_kernel(rw::Args A) { rwb_fwd_body<8>(A); }

__global__ __launch_bounds__(512, 1) void conv_rwb_dgrad_kernel(rw::Args A) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds_h[];
  rwb::build_filter<MODE_DGRAD, 512>(lds_h, A.p[0][0].w, nullptr, threadIdx.x);
  __syncthreads();
  rwb::run_layer<MODE_DGRAD, 8>(A.g[0], A.p[0][0], A.p[0][1], lds_h, blockIdx.x, gridDim.x);
}

// data gradient alone (256-thread workgroups: the form that shares a launch with the weight gradient below)
__global__ __launch_bounds__(256, 2) void conv_rw_dgrad_kernel(rw::Args A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  rw::build_filter<MODE_DGRAD, 256>(lds, A.p[0][0].w, nullptr, threadIdx.x);
  __syncthreads();
  rw::run_layer<MODE_DGRAD, 4>(A.g[0], A.p[0][0], A.p[0][1], lds, blockIdx.x, gridDim.x);
}

#include "conv1_rw.h"

// first layer from a float NHWC minibatch in row-walk form (conv1_rw.h): 512-thread workgroups, one per CU, persistent
// over the samples they own (8 waves share a sample's steps: small minibatches still fill the SIMDs twice)
template <int C>
__global__ __launch_bounds__(512, 2) void conv1_rw_fwd_kernel(rw::Conv1Args a) {
  rw::conv1_body<C, 8>(a, blockIdx.x, gridDim.x);
}

template <int C>
__global__ __launch_bounds__(512, 2) void wgrad1_rw_kernel(rw::Wgrad1Args a) {
  rw::wgrad1_body<C, 8>(a, blockIdx.x, gridDim.x);
}

#include "conv1_u8_rw.h"

// workgroup shape of the uint8 row-walk forward: two 512-thread workgroups per CU (4 waves per SIMD).  Five 256-thread
// ones (5 waves per SIMD, what ~100 VGPRs allow) were slower inside update(): 402 against 407 update()/s
constexpr int kC1U8Threads = 512, kC1U8PerCU = 2;

// first layer straight from the uint8 ring in row-walk form (conv1_u8_rw.h): 512-thread workgroups, two per CU, equal
// shares of the pool of steps of both minibatches (the second with its own weights: both images sit in LDS)
template <int C>
__global__ __launch_bounds__(kC1U8Threads, kC1U8PerCU) void conv1_u8_rw_fwd_kernel(rw::Conv1U8Args a) {
  __shared__ __attribute__((aligned(16))) float lds_img[2 * rw::conv1_u8_image_floats<C>()];
  rw::conv1_u8_stage_weights<C, kC1U8Threads>(lds_img, a.p[0].w, a.p[0].bias, a.scale, threadIdx.x);
  if (a.p[1].B > 0)
    rw::conv1_u8_stage_weights<C, kC1U8Threads>(lds_img + rw::conv1_u8_image_floats<C>(), a.p[1].w, a.p[1].bias, a.scale,
                                                threadIdx.x);
  __syncthreads();
  if ((a.Ws * C) % 4 == 0)
    rw::conv1_u8_body<C, kC1U8Threads / 64, true>(a, lds_img, blockIdx.x, gridDim.x);
  else
    rw::conv1_u8_body<C, kC1U8Threads / 64, false>(a, lds_img, blockIdx.x, gridDim.x);
}

// ... and on the bf16 matrix cores (uint8 pixels are exact in bf16: one part for the pixels, three for the weights; C <= 10).
// The weight fragments take 72 registers (148 in all): three 256-thread workgroups per CU (3 waves per SIMD)
constexpr int kC1U8bThreads = 256, kC1U8bPerCU = 3;
template <int C>
__global__ __launch_bounds__(kC1U8bThreads, kC1U8bPerCU) void conv1_u8_rwb_fwd_kernel(rw::Conv1U8Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_imgb[2 * rw::kConv1U8B3ImageBytes];
  rw::conv1_u8b_stage_weights<C, kC1U8bThreads>(lds_imgb, a.p[0].w, a.p[0].bias, a.scale, threadIdx.x);
  if (a.p[1].B > 0)
    rw::conv1_u8b_stage_weights<C, kC1U8bThreads>(lds_imgb + rw::kConv1U8B3ImageBytes, a.p[1].w, a.p[1].bias, a.scale,
                                                  threadIdx.x);
  __syncthreads();
  if ((a.Ws * C) % 4 == 0)
    rw::conv1_u8b_body<C, kC1U8bThreads / 64, true>(a, lds_imgb, blockIdx.x, gridDim.x);
  else
    rw::conv1_u8b_body<C, kC1U8bThreads / 64, false>(a, lds_imgb, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------
// first layer: Cin = C (9 or 12 ...), stride 2, input either the uint8 replay
// frames (gather by index + random-crop offsets + /255 fused into the load) or
// a float NCHW tensor in [0,255] (the reference's tensor contract).
// K index of the GEMM is k = dy*KR + (dx*C + c), KR = 3C rounded up to 4; the
// (dx,c) run is contiguous in an HWC row, so one LDS row holds it directly.
// ---------------------------------------------------------------------------
struct Conv1Args {
  const void* src;     // SRC_U8: frames [N][Hs][Ws][C] u8;  SRC_F32: [B][C][Hc][Wc] f32
  const int64_t* idx;  // [B] frame index (u8 source) or nullptr -> b
  const int32_t* h1;   // [B] crop row offset or nullptr -> 0
  const int32_t* w1;   // [B] crop col offset or nullptr -> 0
  const float* w;      // OIHW [32][C][3][3]
  const float* bias;   // [32]
  float* out;          // [B][Ho][Wo][32]
  int B, C, Hs, Ws, Hc, Wc, Ho, Wo, th, nbands;
  float scale;
  int dbg;
  // uint8 forward only: a second minibatch from the same ring with its own weights (B2 samples; 0 = none), whose
  // workgroups follow the first one's in the same launch
  const int64_t* idx2;
  const int32_t* h1_2;
  const int32_t* w1_2;
  const float* w2;
  const float* bias2;
  float* out2;
  int B2;
};

enum { SRC_U8 = 0, SRC_F32 = 1, SRC_NHWC = 2 };  // u8 ring / float NCHW tensor / float NHWC tensor

__device__ __forceinline__ int conv1_row_stride(int Wc, int C) { return ((Wc * C + 3) & ~3) + 4; }
__device__ __forceinline__ bool aligned16_dev(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Stage input rows [r0, r0+rows) of sample b's (cropped) image into LDS as
// f32 HWC rows of stride RS, scaled by `scale`.
template <int SRC>
__device__ __forceinline__ void conv1_stage(float* lds, const void* src, const int64_t* idx, const int32_t* h1,
                                            const int32_t* w1, int b, int C, int Hs, int Ws, int Hc, int Wc, int r0,
                                            int rows, int RS, float scale, int tid, int nthreads) {
  const int rowf = Wc * C;
  if (SRC == SRC_U8) {
    const int64_t fi = idx ? idx[b] : b;
    const int oh = h1 ? h1[b] : 0, ow = w1 ? w1[b] : 0;
    const uint8_t* frame = static_cast<const uint8_t*>(src) + (size_t)fi * Hs * Ws * C;
    const int G = (rowf + 3) >> 2;
    for (int i = tid; i < rows * G; i += nthreads) {
      const int r = i / G, g = i - r * G;
      const uint8_t* p = frame + ((size_t)(oh + r0 + r) * Ws + ow) * C + 4 * g;
      // (pointer arithmetic, not an integer round trip: the loads stay global_load, not flat_load)
      const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3);
      const uint32_t* q = reinterpret_cast<const uint32_t*>(p - sh);
      const uint32_t d0 = q[0];
      const uint32_t d1 = sh ? q[1] : 0u;   // only touch the next dword when the run straddles it
      const uint32_t v = __builtin_amdgcn_alignbyte(d1, d0, sh);
      f32x4 o;
      o[0] = (float)(v & 0xff) * scale;
      o[1] = (float)((v >> 8) & 0xff) * scale;
      o[2] = (float)((v >> 16) & 0xff) * scale;
      o[3] = (float)(v >> 24) * scale;
      *reinterpret_cast<f32x4*>(lds + r * RS + 4 * g) = o;
    }
  } else if (SRC == SRC_NHWC) {
    // float NHWC minibatch (augmented observations): rows are contiguous runs of Wc*C floats
    const float* img = static_cast<const float*>(src) + ((size_t)b * Hc + r0) * rowf;
    if ((rowf & 3) == 0 && aligned16_dev(img)) {
      const int G = rowf >> 2;
      for (int i = tid; i < rows * G; i += nthreads) {
        const int r = i / G, g = i - r * G;
        f32x4 v = *reinterpret_cast<const f32x4*>(img + (size_t)r * rowf + 4 * g);
        *reinterpret_cast<f32x4*>(lds + r * RS + 4 * g) = v * scale;
      }
    } else {
      for (int i = tid; i < rows * rowf; i += nthreads) {
        const int r = i / rowf, e = i - r * rowf;
        lds[r * RS + e] = img[(size_t)r * rowf + e] * scale;
      }
    }
  } else {
    const float* img = static_cast<const float*>(src) + (size_t)b * C * Hc * Wc;
    const int n = rows * rowf;
    for (int i = tid; i < n; i += nthreads) {
      const int x = i % Wc;
      const int t = i / Wc;
      const int r = t % rows, c = t / rows;
      lds[r * RS + x * C + c] = img[((size_t)c * Hc + r0 + r) * Wc + x] * scale;
    }
  }
  // The k-steps of a tap row cover KR = 3C rounded up to 4 values: at an odd crop width the last pixel's run ends at
  // the row's end and its padding value is the float BEHIND the row.  Its weight is zero, but 0 x (whatever bit
  // pattern an earlier kernel left in LDS: NaN, Inf) is NaN, which the ReLU then turns into 0 -- a wrong, finite
  // output.  The slack behind every row is zeroed here (the uint8 path wrote whole groups of four: behind those).
#ifndef CURLA_TEST_NO_SLACK_ZERO  // (defined only by a one-off build that checks the regression test can fail)
  {
    const int first = SRC == SRC_U8 ? (rowf + 3) & ~3 : rowf;
    const int pad = RS - first;  // 4..7 floats
    for (int i = tid; i < rows * pad; i += nthreads) {
      const int r = i / pad, e = i - r * pad;
      lds[r * RS + first + e] = 0.f;
    }
  }
#endif
}

template <int SRC, int C>
__global__ __launch_bounds__(512) void conv1_fwd_kernel(Conv1Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KR = (3 * C + 3) & ~3;
  constexpr int NS = 3 * KR / 4;  // k-steps
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int RS = conv1_row_stride(a.Wc, C);

  constexpr int KQ = KR / 4;  // k-steps per tap row
  for (int i = tid; i < 32 * C * 9; i += 512) lds[i] = a.w[i];
  __syncthreads();
  // k = 4 s + kq of the GEMM is (dy, rr) = (s / KQ, 4 (s % KQ) + kq): KR is a multiple of 4, so the tap row is the
  // same for all lanes of a k-step and a lane's operand sits at a COMPILE-TIME offset (dy, 4 (s % KQ)) from its own
  // base (pixel, kq) -- one address per tile instead of one add per k-step (a VALU instruction is a cycle the f32
  // matrix pipe idles: conv_rw.h)
  float wr[NS][2];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int dy = s / KQ, rr = 4 * (s % KQ) + kq;
    const bool ok = rr < 3 * C;
    const int dx = ok ? rr / C : 0, c = ok ? rr - dx * C : 0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) wr[s][mt] = ok ? lds[((mt * 16 + li) * C + c) * 9 + dy * 3 + dx] : 0.f;
  }
  f32x4 bias4[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) bias4[mt] = *reinterpret_cast<const f32x4*>(a.bias + mt * 16 + 4 * kq);
  __syncthreads();

  // persistent: the weight registers above are built once per workgroup, not once per band (a band is ~2 us of
  // tile work at 168x168x12 -- the per-band weight phase was a quarter of the kernel)
  const int nitems = a.B * a.nbands;
  const int qstep = 128 / a.Wo, rstep = 128 - qstep * a.Wo;  // 8 waves x 16 pixels further
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int band = item / a.B, b = item - band * a.B;  // band-major: every workgroup sees every band size
    const int y0 = band * a.th;
    const int tha = min(a.th, a.Ho - y0);
    conv1_stage<SRC>(lds, a.src, a.idx, a.h1, a.w1, b, C, a.Hs, a.Ws, a.Hc, a.Wc, 2 * y0, 2 * tha + 1, RS, a.scale, tid,
                     512);
    __syncthreads();

    const int npix = tha * a.Wo;
    const int ntiles = (npix + 15) >> 4;
    float* const out_item = a.out + ((size_t)(b * a.Ho + y0) * a.Wo) * 32 + 4 * kq;
    int ty = (wave * 16 + li) / a.Wo, x = (wave * 16 + li) - ty * a.Wo;  // walked incrementally: no division per tile
    for (int t = wave; t < ntiles; t += 8) {
      const bool pv = t * 16 + li < npix;
      if (!pv) ty = 0, x = 0;
      const float* base = lds + __mul24(2 * ty, RS) + __mul24(2 * x, C) + kq;
      // the 3 tap rows: RS is a run-time stride, so each row has its own base register; inside a row the k-steps are
      // immediates.  All NS reads of the tile are issued up front (they are independent of the accumulators).
      float bv[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) bv[s] = base[(s / KQ) * RS + 4 * (s % KQ)];
      f32x4 acc[2] = {bias4[0], bias4[1]};  // bias through the accumulators' initial values
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        acc[0] = mfma16(wr[s][0], bv[s], acc[0]);
        acc[1] = mfma16(wr[s][1], bv[s], acc[1]);
      }
      if (pv) {
        float* o = out_item + (__mul24(ty, a.Wo) + x) * 32;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          f32x4 v = acc[mt];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          act_store(reinterpret_cast<f32x4*>(o + mt * 16), v);
        }
      }
      x += rstep, ty += qstep;  // at most one more wrap
      const bool wrap = x >= a.Wo;
      x = wrap ? x - a.Wo : x;
      ty = wrap ? ty + 1 : ty;
    }
    __syncthreads();  // every wave is done with the band before the next one is staged over it
  }
}

// ---------------------------------------------------------------------------
// first layer, uint8 ring source, bytes kept as bytes in LDS (4x less LDS than the float band: a whole
// 76x76x9 crop is 52 KB, so a workgroup takes one sample with no halo re-reads and two workgroups share a
// CU).  Staging is a pure byte copy: 16-byte runs of the (arbitrarily aligned) crop row are rebuilt from
// aligned dword loads with v_alignbyte; u8 -> f32 and the /255 happen when the MFMA B operand is read.
// k = 4s + kq of the GEMM is (dy, rr) = (s / (KR/4), 4 (s % (KR/4)) + kq): since KR is a multiple of 4 the
// tap row dy is wave-uniform per k-step and the byte offset inside the row differs per lane only by kq.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int conv1_row_bytes(int Wc, int C) { return ((Wc * C + 15) & ~15) + 16; }

// One pass of the byte staging, split into its two halves so that a kernel can put other work between the
// loads and the LDS stores: U 16-byte runs per thread, all their dword loads in flight together.
template <int U>
struct Conv1StageRegs {
  uint32_t dw[U][5], sh[U];
  int dst[U];
};

template <int U>
__device__ __forceinline__ void conv1_stage_u8_issue(Conv1StageRegs<U>& rg, const uint8_t* frame, int oh, int ow, int C,
                                                     int Ws, int Wc, int r0, int rows, int RSb, int i0, int tid,
                                                     int nthreads) {
  const int runs = (Wc * C + 15) >> 4;  // 16-byte runs per row
  const int total = rows * runs;
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const int i = i0 + tid + k * nthreads;
    const bool ok = i < total;
    const int ic = ok ? i : 0;
    const int r = ic / runs, g = ic - r * runs;
    const uint8_t* p = frame + ((size_t)(oh + r0 + r) * Ws + ow) * C + 16 * g;
    // (pointer arithmetic, not an integer round trip: the loads stay global_load, not flat_load)
    rg.sh[k] = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p - rg.sh[k]);
    rg.dst[k] = ok ? r * RSb + 16 * g : -1;
#pragma unroll
    for (int e = 0; e < 4; ++e) rg.dw[k][e] = ok ? q[e] : 0u;
    rg.dw[k][4] = (ok && rg.sh[k]) ? q[4] : 0u;  // only touch the fifth dword when the run straddles it
  }
}

template <int U>
__device__ __forceinline__ void conv1_stage_u8_commit(const Conv1StageRegs<U>& rg, uint8_t* lds) {
#pragma unroll
  for (int k = 0; k < U; ++k) {
    uint4 o;
    o.x = __builtin_amdgcn_alignbyte(rg.dw[k][1], rg.dw[k][0], rg.sh[k]);
    o.y = __builtin_amdgcn_alignbyte(rg.dw[k][2], rg.dw[k][1], rg.sh[k]);
    o.z = __builtin_amdgcn_alignbyte(rg.dw[k][3], rg.dw[k][2], rg.sh[k]);
    o.w = __builtin_amdgcn_alignbyte(rg.dw[k][4], rg.dw[k][3], rg.sh[k]);
    if (rg.dst[k] >= 0) *reinterpret_cast<uint4*>(lds + rg.dst[k]) = o;
  }
}

__device__ __forceinline__ void conv1_stage_u8(uint8_t* lds, const uint8_t* frames, const int64_t* idx,
                                               const int32_t* h1, const int32_t* w1, int b, int C, int Hs, int Ws,
                                               int Wc, int r0, int rows, int RSb, int tid, int nthreads,
                                               int first_run = 0) {
  const int64_t fi = idx ? idx[b] : b;
  const int oh = h1 ? h1[b] : 0, ow = w1 ? w1[b] : 0;
  const uint8_t* frame = frames + (size_t)fi * Hs * Ws * C;
  const int total = rows * ((Wc * C + 15) >> 4);
  constexpr int U = 4;
  for (int i0 = first_run; i0 < total; i0 += nthreads * U) {
    Conv1StageRegs<U> rg;
    conv1_stage_u8_issue<U>(rg, frame, oh, ow, C, Ws, Wc, r0, rows, RSb, i0, tid, nthreads);
    conv1_stage_u8_commit<U>(rg, lds);
  }
}

// The same byte staging dealt out BY ROW: wave w of NW takes crop rows w, w + NW, ..., lane g the row's g-th 16-byte
// run (lanes past the row idle).  A row's address, its misalignment and its LDS offset are then wave-uniform -- scalar
// registers and scalar arithmetic -- and a run costs a lane one address add instead of an integer division by the run
// count and 64-bit pointer arithmetic (the element-per-thread form above: ~40 VALU instructions per run, ~3800 wave
// instructions per 76x76x9 crop against the 2700 of the multiply loop that follows).  UR rows per wave and call.
template <int UR>
struct Conv1RowRegs {
  uint32_t dw[UR][5];
  uint32_t sh[UR];  // (wave-uniform)
  int dst[UR];      // (wave-uniform row offset; < 0: no row)
};

template <int UR>
__device__ __forceinline__ void conv1_stage_rows_issue(Conv1RowRegs<UR>& rg, const uint8_t* crop, int pitch, int nbytes,
                                                       int rows, int RSb, int r_first, int wave, int nwaves, int lane) {
  const bool lane_on = 16 * lane < nbytes;
#pragma unroll
  for (int k = 0; k < UR; ++k) {
    const int r = r_first + wave + k * nwaves;  // (uniform)
    rg.dst[k] = r < rows ? r * RSb : -1;
    const uint8_t* p = crop + (size_t)min(r, rows - 1) * pitch;
    rg.sh[k] = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p - rg.sh[k]) + 4 * lane;
    // (registers of rows / lanes that load nothing stay undefined: the commit never stores them)
    if (r < rows && lane_on) {
#pragma unroll
      for (int e = 0; e < 4; ++e) rg.dw[k][e] = q[e];
      if (rg.sh[k]) rg.dw[k][4] = q[4];  // only touch the fifth dword when the run straddles it
    }
  }
}

template <int UR>
__device__ __forceinline__ void conv1_stage_rows_commit(const Conv1RowRegs<UR>& rg, uint8_t* lds, int nbytes, int lane) {
  const bool lane_on = 16 * lane < nbytes;
#pragma unroll
  for (int k = 0; k < UR; ++k) {
    uint4 o;
    o.x = __builtin_amdgcn_alignbyte(rg.dw[k][1], rg.dw[k][0], rg.sh[k]);
    o.y = __builtin_amdgcn_alignbyte(rg.dw[k][2], rg.dw[k][1], rg.sh[k]);
    o.z = __builtin_amdgcn_alignbyte(rg.dw[k][3], rg.dw[k][2], rg.sh[k]);
    o.w = __builtin_amdgcn_alignbyte(rg.dw[k][4], rg.dw[k][3], rg.sh[k]);
    if (rg.dst[k] >= 0 && lane_on) *reinterpret_cast<uint4*>(lds + rg.dst[k] + 16 * lane) = o;
  }
}

// all rows [r_first, rows) of a crop, UR per wave and pass
template <int UR>
__device__ __forceinline__ void conv1_stage_rows(uint8_t* lds, const uint8_t* crop, int pitch, int nbytes, int rows,
                                                 int RSb, int r_first, int wave, int nwaves, int lane) {
  for (int r0 = r_first; r0 < rows; r0 += UR * nwaves) {
    Conv1RowRegs<UR> rg;
    conv1_stage_rows_issue<UR>(rg, crop, pitch, nbytes, rows, RSb, r0, wave, nwaves, lane);
    conv1_stage_rows_commit<UR>(rg, lds, nbytes, lane);
  }
}

// float4 copy of `n4` contiguous float4 from HBM into the pixel-padded LDS band layout (8 float4 per pixel ->
// stride kLdsPix floats), U loads in flight per thread per pass
__device__ __forceinline__ void stage_band_f32(float* lds_band, const float* src, int n4, int tid, int nthreads) {
  constexpr int U = 6;
  for (int f0 = 0; f0 < n4; f0 += nthreads * U) {
    f32x4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int f = f0 + tid + k * nthreads;
      v[k] = f < n4 ? *reinterpret_cast<const f32x4*>(src + (size_t)f * 4) : f32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int f = f0 + tid + k * nthreads;
      if (f < n4) *reinterpret_cast<f32x4*>(lds_band + (f >> 3) * kLdsPix + (f & 7) * 4) = v[k];
    }
  }
}

template <int C>
__global__ __launch_bounds__(512) void conv1_fwd_u8_kernel(Conv1Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KR = (3 * C + 3) & ~3;
  constexpr int KQ = KR / 4;      // k-steps per tap row
  constexpr int NS = 3 * KQ;      // k-steps
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int RSb = conv1_row_bytes(a.Wc, C);

  // (two problems in one launch: the workgroups of the second minibatch follow the first's and take its weights)
  const int nitems1 = a.B * a.nbands;
  const bool second = (int)blockIdx.x >= nitems1;
  const int item = second ? blockIdx.x - nitems1 : blockIdx.x;
  const int Bc = second ? a.B2 : a.B;
  if (second) a.idx = a.idx2, a.h1 = a.h1_2, a.w1 = a.w1_2, a.w = a.w2, a.bias = a.bias2, a.out = a.out2;
  const int band = item / Bc, b = item - band * Bc;
  const int y0 = band * a.th;
  const int tha = min(a.th, a.Ho - y0);
  // the first staging pass (7 x 512 runs: a whole 76x76x9 crop) is issued before the weight phase, whose two
  // barriers and LDS gather then run under the loads' latency; its LDS stores come after (same LDS region)
  constexpr int U0 = 7;
  Conv1StageRegs<U0> rg0;
  const uint8_t* frame0;
  int oh0, ow0;
  {
    const int64_t fi = a.idx ? a.idx[b] : b;
    oh0 = a.h1 ? a.h1[b] : 0, ow0 = a.w1 ? a.w1[b] : 0;
    frame0 = static_cast<const uint8_t*>(a.src) + (size_t)fi * a.Hs * a.Ws * C;
    if (!ABL(1)) conv1_stage_u8_issue<U0>(rg0, frame0, oh0, ow0, C, a.Ws, a.Wc, 2 * y0, 2 * tha + 1, RSb, 0, tid, 512);
  }

  // weights -> MFMA A-operand registers through a k-major LDS image [dy][rr (KR, zero padded)][cout 32], rr = dx*C + c,
  // with the 1/255 of `obs / 255.` (encoder.py:78) folded in: the index arithmetic is paid once per weight while
  // staging (5 per thread), and every lane then reads its 2 x NS values at compile-time offsets from ONE base
  // (the per-register gather out of the OIHW image cost ~10 VALU for each of the 42 registers of every lane).
  for (int i = tid; i < 3 * KR * 32; i += 512) {
    const int co = i & 31, k = i >> 5;
    const int dy = k / KR, rr = k - dy * KR;
    const int dx = rr / C, c = rr - dx * C;
    lds[i] = rr < 3 * C ? a.w[(co * C + c) * 9 + dy * 3 + dx] * a.scale : 0.f;
  }
  __syncthreads();
  float wr[NS][2];
  {
    const float* wl = lds + kq * 32 + li;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) wr[s][mt] = wl[((s / KQ) * KR + 4 * (s % KQ)) * 32 + mt * 16];
  }
  f32x4 bias4[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) bias4[mt] = *reinterpret_cast<const f32x4*>(a.bias + mt * 16 + 4 * kq);
  __syncthreads();

  uint8_t* ldsb = reinterpret_cast<uint8_t*>(lds);
  if (!ABL(1)) {
    conv1_stage_u8_commit<U0>(rg0, ldsb);
    conv1_stage_u8(ldsb, static_cast<const uint8_t*>(a.src), a.idx, a.h1, a.w1, b, C, a.Hs, a.Ws, a.Wc, 2 * y0,
                   2 * tha + 1, RSb, tid, 512, /*first_run=*/U0 * 512);  // taller bands: the rest
  }
  __syncthreads();

  const int npix = tha * a.Wo;
  const int ntiles = ABL(64) ? 0 : (npix + 15) >> 4;
  int ty = (wave * 16 + li) / a.Wo, x = (wave * 16 + li) - ty * a.Wo;
  const int qstep = 128 / a.Wo, rstep = 128 - qstep * a.Wo;
  for (int t = wave; t < ntiles; t += 8) {
    const bool pv = t * 16 + li < npix;
    if (!pv) ty = 0, x = 0;
    const uint8_t* base = ldsb + 2 * ty * RSb + 2 * x * C + kq;
    // all NS byte reads of the tile are issued first; each conversion is then placed one k-step ahead of the MFMA
    // pair that consumes it (left alone, the compiler emits read -> wait -> convert -> s_nop -> 2 MFMAs chains that
    // expose the LDS latency and a VALU->MFMA hazard stall on every k-step: tools/micro/conv1_loop.hip)
    uint32_t raw[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) raw[s] = base[(s / KQ) * RSb + 4 * (s % KQ)];
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[2] = {bias4[0], bias4[1]};  // bias through the accumulators' initial values
    // (the conversions are volatile asm so that instruction selection cannot sink them next to their users; the
    // first one carries the 2 wait states a VALU write needs before an MFMA reads it, every other one has the two
    // MFMAs of the previous k-step between itself and its reader: tools/check_asm_hazards.py scans the ISA)
    float cur;
    asm volatile("v_cvt_f32_ubyte0 %0, %1\n\ts_nop 1" : "=v"(cur) : "v"(raw[0]));
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float nxt = cur;
      if (s + 1 < NS) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(nxt) : "v"(raw[s + 1]));
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = mfma16(wr[s][0], cur, acc[0]);
      acc[1] = mfma16(wr[s][1], cur, acc[1]);
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
    if (pv && !ABL(4)) {
      const size_t g = ((size_t)(b * a.Ho + y0 + ty) * a.Wo + x) * 32 + 4 * kq;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 v = acc[mt];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        act_store(reinterpret_cast<f32x4*>(a.out + g + mt * 16), v);
      }
    }
    x += rstep, ty += qstep;  // 8 waves x 16 pixels further: qstep rows + rstep columns, at most one more wrap
    const bool wrap = x >= a.Wo;
    x = wrap ? x - a.Wo : x;
    ty = wrap ? ty + 1 : ty;
  }
}

// ---------------------------------------------------------------------------
// The same layer, HYBRID form: the banded kernel's input side (one workgroup per sample, the crop staged as bytes in
// LDS by one burst of aligned loads -- which is what makes that kernel indifferent to where the ring slots come from)
// with the row walk's compute loop (conv1_u8_rw.h: a wave owns 16 output columns and walks down; a lane group's
// E = ceil(3C/4) operand bytes of an input row are CONTIGUOUS, here read from LDS as aligned dwords + v_alignbyte, two
// new rows per 6 E MFMAs) instead of one byte read + one (row, column) walk per k-step.  The sample's steps (strips x
// rows, rw::Geom) are split evenly over the 8 waves.  Only for crops that fit one band (nbands == 1).
// ---------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(512, 2) void conv1_u8_walk_kernel(Conv1Args a, rw::Geom G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int E = (3 * C + 3) / 4, KR = 4 * E;
  constexpr int NLD = (E + 3 + 3) / 4;  // aligned dwords that hold a run starting at byte 0..3 of the first
  constexpr int NWD = (E + 3) / 4;      // dwords of the run once it starts at byte 0
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int RSb = conv1_row_bytes(a.Wc, C);
  const bool second = (int)blockIdx.x >= a.B;
  const int b = second ? blockIdx.x - a.B : blockIdx.x;
  if (second) a.idx = a.idx2, a.h1 = a.h1_2, a.w1 = a.w1_2, a.w = a.w2, a.bias = a.bias2, a.out = a.out2;
  // the crop's bytes: requested before the weight phase, stored to LDS after it (same region)
  constexpr int U0 = 10;  // rows per wave in flight across the weight phase (8 waves x 10 >= the 77 rows of a 76x76 crop)
  Conv1RowRegs<U0> rg0;
  const int crop_rows = 2 * a.Ho + 1, crop_bytes = a.Wc * C;
  const uint8_t* crop0;
  {
    const int64_t fi = a.idx ? rw::const_load(a.idx, b) : (int64_t)b;  // (scalar loads: b is wave-uniform)
    const int oh0 = a.h1 ? rw::const_load(a.h1, b) : 0, ow0 = a.w1 ? rw::const_load(a.w1, b) : 0;
    crop0 = static_cast<const uint8_t*>(a.src) + ((size_t)fi * a.Hs + oh0) * a.Ws * C + (size_t)ow0 * C;
  }
  conv1_stage_rows_issue<U0>(rg0, crop0, a.Ws * C, crop_bytes, crop_rows, RSb, 0, wave, 8, lane);  // (host: <= 64 runs per row)
  for (int i = tid; i < 3 * KR * 32; i += 512) {
    const int co = i & 31, k = i >> 5;
    const int dy = k / KR, rr = k - dy * KR;
    const int dx = rr / C, c = rr - dx * C;
    lds[i] = rr < 3 * C ? a.w[(co * C + c) * 9 + dy * 3 + dx] * a.scale : 0.f;
  }
  __syncthreads();
  float wr[3][E][2];  // lane (li = cout, kq): W[cout][dy][rr = E kq + e] * scale
  {
    const float* wl = lds + (E * kq) * 32 + li;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int e = 0; e < E; ++e)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) wr[dy][e][mt] = wl[(dy * KR + e) * 32 + mt * 16];
  }
  f32x4 bias4[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) bias4[mt] = *reinterpret_cast<const f32x4*>(a.bias + mt * 16 + 4 * kq);
  __syncthreads();
  uint8_t* ldsb = reinterpret_cast<uint8_t*>(lds);
  conv1_stage_rows_commit<U0>(rg0, ldsb, crop_bytes, lane);
  conv1_stage_rows<4>(ldsb, crop0, a.Ws * C, crop_bytes, crop_rows, RSb, /*r_first=*/U0 * 8, wave, 8, lane);
  __syncthreads();

  const int lo = G.steps * wave / 8, hi = G.steps * (wave + 1) / 8;
  const int out_row = a.Wo * 128;
  const __amdgpu_buffer_rsrc_t rout = rw::uniform_rsrc(a.out + (size_t)b * a.Ho * a.Wo * 32, a.Ho * out_row);
  for (int g = lo; g < hi;) {
    int k, sb, n_strip;
    if (g < G.nfull * G.Ho) {
      k = g / G.Ho, sb = g - k * G.Ho, n_strip = G.Ho;
    } else {
      const int q = (g - G.nfull * G.Ho) / G.nr;
      k = G.nfull + q, sb = g - G.nfull * G.Ho - q * G.nr, n_strip = G.nr;
    }
    const int n = hi - g < n_strip - sb ? hi - g : n_strip - sb;
    g += n;
    int x, y0;
    bool lane_on;
    if (k < G.nfull) {
      x = 16 * k + li, y0 = 0, lane_on = true;
    } else {
      const int u = (k - G.nfull) * 16 + li;
      const int col = u / G.nseg, sg = u - col * G.nseg;
      lane_on = col < G.brem;
      x = 16 * G.nfull + col, y0 = sg * G.nr;
    }
    // (lanes without a column, and rows past the crop, read whatever sits in LDS: finite bytes; nothing of it is stored)
    const int Y = lane_on ? min(y0 + sb, a.Ho - 1) : 0;
    const int xx = lane_on ? x : 0;
    const unsigned run = (unsigned)(2 * Y * RSb + 2 * xx * C + E * kq);
    const unsigned sh = run & 3u;
    const uint8_t* rowp = ldsb + (run & ~3u);
    unsigned vo = lane_on ? (unsigned)(((y0 + sb) * a.Wo + x) * 128 + kq * 16) : 0x80000000u;
    const int rmax = 2 * a.Ho - 2 * Y;  // last crop row (relative to 2 Y) that exists in LDS
    struct Raw {
      uint32_t d[NLD];
    };
    struct Row {
      float v[E];
    };
    auto load_row = [&](Raw& R, int r) {  // crop row 2 Y + r (clamped into the staged image)
      const uint32_t* p = reinterpret_cast<const uint32_t*>(rowp + __mul24(min(r, rmax), RSb));  // (24-bit: no 64-bit mad)
#pragma unroll
      for (int j = 0; j < NLD; ++j) R.d[j] = p[j];
    };
    auto convert = [&](Row& F, const Raw& R) {
      rw::RawBytes<NWD> Wd;
#pragma unroll
      for (int j = 0; j < NWD; ++j) Wd.d[j] = __builtin_amdgcn_alignbyte(j + 1 < NLD ? R.d[j + 1 < NLD ? j + 1 : j] : 0u, R.d[j], sh);
#pragma unroll
      for (int e = 0; e < E; ++e) F.v[e] = rw::byte_f32<NWD>(Wd, e);
    };
    auto mma_row = [&](f32x4 (&acc)[2], const Row& F, const int dy) {
#pragma unroll
      for (int e = 0; e < E; ++e)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma16(wr[dy][e][mt], F.v[e], acc[mt]);
    };
    struct Pair {
      Raw a, b;
    };
    auto step = [&](const Row& r0, Row& r1, Row& r2, const Pair& cur, Pair& nxt, const int t) {
      load_row(nxt.a, 2 * t + 3);
      load_row(nxt.b, 2 * t + 4);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc[2] = {bias4[0], bias4[1]};
      mma_row(acc, r0, 0);
      __builtin_amdgcn_sched_barrier(0);
      convert(r1, cur.a);
      convert(r2, cur.b);
      __builtin_amdgcn_sched_barrier(0);
      mma_row(acc, r1, 1);
      mma_row(acc, r2, 2);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 v = acc[mt];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rw::relu_bits(v[r]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rout,
                                               vo + mt * 64u, 0, CURLA_ACT_STORE_POLICY);
      }
      vo += out_row;
    };
    Row S0, S1, S2, S3, S4;
    Pair P0, P1;
    {
      Raw R0;
      load_row(R0, 0), load_row(P0.a, 1), load_row(P0.b, 2);
      convert(S0, R0);
    }
    for (int t = 0;;) {  // rows of step t sit in sets (2t, 2t+1, 2t+2) mod 5, its bytes in pair t mod 2
      step(S0, S1, S2, P0, P1, t);
      if (++t >= n) break;
      step(S2, S3, S4, P1, P0, t);
      if (++t >= n) break;
      step(S4, S0, S1, P0, P1, t);
      if (++t >= n) break;
      step(S1, S2, S3, P1, P0, t);
      if (++t >= n) break;
      step(S3, S4, S0, P0, P1, t);
      if (++t >= n) break;
      step(S0, S1, S2, P1, P0, t);
      if (++t >= n) break;
      step(S2, S3, S4, P0, P1, t);
      if (++t >= n) break;
      step(S4, S0, S1, P1, P0, t);
      if (++t >= n) break;
      step(S1, S2, S3, P0, P1, t);
      if (++t >= n) break;
      step(S3, S4, S0, P1, P0, t);
      if (++t >= n) break;
    }
  }
}

// ---------------------------------------------------------------------------
// weight gradient, stride-1 32->32:  dW[co][ci][tap] = sum_pixels g[p][co] * in[p+tap][ci]  (conv_rw_wgrad.h).
// Partial sums leave through one slab of kPartialS1 floats per workgroup and a deterministic second pass
// (wgrad_reduce_multi_kernel below).
// ---------------------------------------------------------------------------
constexpr int kPartialS1 = 32 * 288 + 32;

#include "conv_rw_wgrad.h"
#include "conv_rw_wgrad2.h"

// the weight-gradient body of a workgroup: Winograd along x (conv_rw_wgrad.h) or in both directions (conv_rw_wgrad2.h)
__device__ __forceinline__ void wgrad_any(const rw::WgradArgs& wa, const int bid, const int nblk) {
  if (wa.two_d)
    rw::wgrad2_body<4>(wa, bid, nblk);
  else
    rw::wgrad_body<4>(wa, bid, nblk);
}

// weight gradient in its row-walk form (conv_rw_wgrad.h), alone and in one launch with the row-walk data gradient
__global__ __launch_bounds__(256, 2) void wgrad_rw_kernel(rw::WgradArgs wa) {
  wgrad_any(wa, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256, 2) void bwd_rw2_kernel(rw::WgradArgs wa, rw::Args da, int nw) {
  if ((int)blockIdx.x < nw) {
    wgrad_any(wa, blockIdx.x, nw);
  } else {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    rw::build_filter<MODE_DGRAD, 256>(lds, da.p[0][0].w, nullptr, threadIdx.x);
    __syncthreads();
    rw::run_layer<MODE_DGRAD, 4>(da.g[0], da.p[0][0], da.p[0][1], lds, (int)blockIdx.x - nw, (int)gridDim.x - nw);
  }
}

// ... and with the data gradient in its bf16x3 form (conv_rwb.h) beside it: 256-thread workgroups of both kinds, the data
// gradient's with 72 KB of LDS for its split filter
__global__ __launch_bounds__(256, 2) void bwd_rwb2_kernel(rw::WgradArgs wa, rw::Args da, int nw) {
  if ((int)blockIdx.x < nw) {
    wgrad_any(wa, blockIdx.x, nw);
  } else {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds_hb[];
    rwb::build_filter<MODE_DGRAD, 256>(lds_hb, da.p[0][0].w, nullptr, threadIdx.x);
    __syncthreads();
    rwb::run_layer<MODE_DGRAD, 4>(da.g[0], da.p[0][0], da.p[0][1], lds_hb, (int)blockIdx.x - nw, (int)gridDim.x - nw);
  }
}

// ---------------------------------------------------------------------------
// weight gradient of the first layer (stride 2, Cin = C, input re-read from
// the uint8 frames / float tensor exactly as the forward does).
// D[co][k'] with k' = dy*KR + dx*C + c (the forward's K index), K = pixels.
// ---------------------------------------------------------------------------
struct Wgrad1Args {
  const void* src;
  const int64_t* idx;
  const int32_t* h1;
  const int32_t* w1;
  const float* g;  // [B][Ho][Wo][32]
  float* partial;  // [grid][32*C*9 + 32]
  int B, C, Hs, Ws, Hc, Wc, Ho, Wo, th, nbands;
  float scale;
  unsigned lds_bytes;  // dynamic LDS of the launch (the uint8 kernel sizes its final cross-wave sum by it)
  int dbg;
};

template <int SRC, int C>
__global__ __launch_bounds__(512) void wgrad1_kernel(Wgrad1Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KR = (3 * C + 3) & ~3;
  constexpr int NT = (3 * KR + 15) / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int RS = conv1_row_stride(a.Wc, C);
  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0, 0, 0, 0};
  float bsum[2] = {0.f, 0.f};
  int koff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int k = t * 16 + li;
    const int dy = k / KR, rr = k - dy * KR;
    koff[t] = (dy < 3) ? dy * RS + rr : 0;
  }

  const int nitems = a.B * a.nbands;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int band = item / a.B, b = item - band * a.B;  // band-major: every workgroup sees every band size
    const int y0 = band * a.th;
    const int tha = min(a.th, a.Ho - y0);
    conv1_stage<SRC>(lds, a.src, a.idx, a.h1, a.w1, b, C, a.Hs, a.Ws, a.Hc, a.Wc, 2 * y0, 2 * tha + 1, RS, a.scale,
                     tid, 512);
    __syncthreads();
    // (the gradient operand is loaded from HBM/L2 straight into MFMA registers, as in wgrad1_u8_kernel: every pixel
    // is needed by exactly one wave, and lanes with nothing to multiply read a zero page)
    const float* const gband = a.g + ((size_t)(b * a.Ho + y0) * a.Wo) * 32 + li;
    const int npix = tha * a.Wo;
    const int nunits = ((npix + 15) >> 4) << 2;
    // unit u's pixel of lane group kq is p = (u >> 2) * 16 + (u & 3) + 4 kq; this wave's units are u = wave, wave + 8,
    // ...: p advances by 32 per unit -- walked incrementally as (row, column), no division per unit
    int p = (wave >> 2) * 16 + (wave & 3) + 4 * kq;
    int ty = p / a.Wo, x = p - ty * a.Wo;
    const int qstep = 32 / a.Wo, rstep = 32 - qstep * a.Wo;
    for (int u = wave; u < nunits; u += 8) {
      const bool pv = p < npix;
      const float* gp = pv ? gband + p * 32 : g_zero_px;
      const float a0 = gp[0], a1 = gp[16];
      bsum[0] += a0;
      bsum[1] += a1;
      const float* ip = lds + (pv ? __mul24(2 * ty, RS) + __mul24(2 * x, C) : 0);
      float bv[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[t] = ip[koff[t]];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[0][t] = mfma16(a0, bv[t], acc[0][t]);
        acc[1][t] = mfma16(a1, bv[t], acc[1][t]);
      }
      p += 32, x += rstep, ty += qstep;  // at most one more wrap
      const bool wrap = x >= a.Wo;
      x = wrap ? x - a.Wo : x;
      ty = wrap ? ty + 1 : ty;
    }
    __syncthreads();
  }

  bsum[0] += __shfl_xor(bsum[0], 16);
  bsum[0] += __shfl_xor(bsum[0], 32);
  bsum[1] += __shfl_xor(bsum[1], 16);
  bsum[1] += __shfl_xor(bsum[1], 32);
  const int nw = 32 * C * 9;
  for (int w = 0; w < 8; ++w) {
    if (wave == w) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int k = t * 16 + li;
          const int dy = k / KR, rr = k - dy * KR;
          if (dy < 3 && rr < 3 * C) {
            const int dx = rr / C, c = rr - dx * C;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int co = mt * 16 + 4 * kq + r;
              float* d = lds + (co * C + c) * 9 + dy * 3 + dx;
              *d = (w == 0) ? acc[mt][t][r] : *d + acc[mt][t][r];
            }
          }
        }
      if (kq == 0) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          float* d = lds + nw + mt * 16 + li;
          *d = (w == 0) ? bsum[mt] : *d + bsum[mt];
        }
      }
    }
    __syncthreads();
  }
  float* slab = a.partial + (size_t)blockIdx.x * (nw + 32);
  for (int i = tid; i < nw + 32; i += 512) slab[i] = lds[i];
}

// first-layer weight gradient from the uint8 ring with the input band kept as bytes in LDS (see
// conv1_fwd_u8_kernel).  The gradient operand never enters LDS: lane (li, kq) of a k-step needs channels li and
// 16 + li of ONE pixel, every pixel is needed by exactly one wave, and the 16 lanes of a group read 64 contiguous
// bytes -- so each wave loads its own operand values from HBM/L2 one k-step ahead (the other three waves of the SIMD
// cover the latency).  That removes three quarters of the staging volume (53 KB of gradients per 16 KB of bytes at
// 84x84x9) and lets a workgroup take a whole crop as bytes.
// NW waves per workgroup: 8 (two workgroups per CU) or 4 (four smaller ones: the stage -> barrier -> multiply ->
// barrier phases of a workgroup do not overlap each other, so what covers a workgroup's staging is the number of
// OTHER workgroups on its CU that are multiplying at that moment)
template <int C, int NW>
__global__ __launch_bounds__(64 * NW) void wgrad1_u8_kernel(Wgrad1Args a) {
  constexpr int NTHR = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // k' = dy * 3C + dx * C + c, unpadded (9C values): NT tiles of 16.  At C = 9 that is 81 = 5 tiles + ONE column; a
  // sixth tile for it would be a sixth of all MFMAs, so that column (dy = 2, dx = 2, c = C-1) is accumulated by two
  // VALU FMAs per k-step instead (lane (li, kq) holds the gradient of channels li / 16+li at its pixel anyway).
  constexpr int K9 = 9 * C;
  constexpr bool TAIL = (K9 % 16 == 1);
  constexpr int NT = TAIL ? K9 / 16 : (K9 + 15) / 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: the piece bookkeeping below stays scalar)
  const int li = lane & 15, kq = lane >> 4;
  const int RSb = conv1_row_bytes(a.Wc, C);
  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0, 0, 0, 0};
  f32x2 bsum = {0.f, 0.f};   // (pairs: one packed add / fma for both halves of the output channels)
  f32x2 atail = {0.f, 0.f};
  int koff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int k = t * 16 + li;
    const int dy = k / (3 * C), rr = k - dy * (3 * C);
    koff[t] = (dy < 3) ? dy * RSb + rr : 0;
  }
  const int koff_tail = 2 * RSb + 3 * C - 1;
  const int nitems = a.B * a.nbands;
  uint8_t* ldsb = reinterpret_cast<uint8_t*>(lds);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int band = item / a.B, b = item - band * a.B;
    const int y0 = band * a.th;
    const int tha = min(a.th, a.Ho - y0);
    if (!ABL(1)) {
      const int64_t fi = a.idx ? rw::const_load(a.idx, b) : (int64_t)b;  // (scalar loads: b is wave-uniform)
      const int oh = a.h1 ? rw::const_load(a.h1, b) : 0, ow = a.w1 ? rw::const_load(a.w1, b) : 0;
      const uint8_t* crop = static_cast<const uint8_t*>(a.src) + ((size_t)fi * a.Hs + oh + 2 * y0) * a.Ws * C + (size_t)ow * C;
      if (a.Wc * C <= 64 * 16)  // a lane per 16-byte run of a row
        conv1_stage_rows<10>(ldsb, crop, a.Ws * C, a.Wc * C, 2 * tha + 1, RSb, 0, wave, NW, lane);
      else
        conv1_stage_u8(ldsb, static_cast<const uint8_t*>(a.src), a.idx, a.h1, a.w1, b, C, a.Hs, a.Ws, a.Wc, 2 * y0,
                       2 * tha + 1, RSb, tid, NTHR);
    }
    __syncthreads();
    const float* const gband = a.g + ((size_t)(b * a.Ho + y0) * a.Wo) * 32 + li;  // the band's pixels are contiguous
    // The walk.  A k-step ("unit") is 4 pixels, lane group kq takes one of them; units come in PIECES of UNR, and the
    // band's pieces are dealt to the waves round-robin (piece w, w + NW, ...).  Two kinds of piece:
    //   row piece:    UNR consecutive units of one output row (pixels 4u + kq): only the floor(Wo / 4) WHOLE units of
    //                 a row, so that no unit multiplies fewer than 4 pixels;
    //   column piece: the Wo % 4 columns a row's whole units leave over, walked DOWN -- a unit is rows 4u + kq of one
    //                 such column (at Wo = 37: 10 units for column 36 instead of a quarter-filled tenth unit in each
    //                 of the 37 rows: 343 units per crop, not 370).
    // In a row piece everything a unit needs sits at a COMPILE-TIME offset from per-piece bases: the byte operands at
    // (piece base + koff[t]) + 8 C jj, the two gradient values at piece base + 128 jj floats -- no per-unit address
    // arithmetic (the pixel-order walk spent ~45 VALU instructions per 12 MFMAs on it: matrix pipe busy 46 %).  Pieces
    // of 3 (not whole rows) so that the waves' shares differ by at most one piece: 115 pieces over 8 waves = 15 each
    // at most, 45 units, where whole rows gave 5 rows x 10 units.  The bytes are multiplied unscaled, `scale` is
    // applied once to the accumulated sums.
    constexpr int UNR = 3;
    const int fullu = a.Wo >> 2, remc = a.Wo & 3;
    const int cpr = (fullu + UNR - 1) / UNR;  // row pieces per row
    const int cpc = (((tha + 3) >> 2) + UNR - 1) / UNR;  // column pieces per left-over column
    const int nrow = tha * cpr, npieces = nrow + remc * cpc;
    const int kdiv = NW / max(cpr, 1), kmod = NW - kdiv * cpr;
    struct Piece {  // (wave-uniform)
      int q, ty, c;  // row piece: row ty, piece c of the row;  column piece (q >= nrow): left-over column ty, piece c
    };
    auto column_of = [&](Piece& p) {
      const int qq = p.q - nrow;
      p.ty = qq / cpc, p.c = qq - p.ty * cpc;
    };
    auto next_piece = [&](Piece& p) {
      p.q += NW;
      if (p.q < nrow) {
        p.c += kmod, p.ty += kdiv;
        if (p.c >= cpr) p.c -= cpr, ++p.ty;
      } else {
        column_of(p);
      }
    };
    // gradient values of a piece (from HBM/L2, one piece ahead of their use; pixels past the band: zero page)
    auto gload = [&](const Piece& p, f32x2 (&av)[UNR]) {
      if (p.q < nrow) {
        const float* gp = gband + (p.ty * a.Wo + 4 * UNR * p.c + kq) * 32;
        if (UNR * (p.c + 1) <= fullu) {
#pragma unroll
          for (int jj = 0; jj < UNR; ++jj) av[jj] = f32x2{gp[128 * jj], gp[128 * jj + 16]};
        } else {
#pragma unroll
          for (int jj = 0; jj < UNR; ++jj) {
            const float* g1 = UNR * p.c + jj < fullu ? gp + 128 * jj : g_zero_px;
            av[jj] = f32x2{g1[0], g1[16]};
          }
        }
      } else {
        const int x = 4 * fullu + p.ty;
#pragma unroll
        for (int jj = 0; jj < UNR; ++jj) {
          const int y = 4 * (UNR * p.c + jj) + kq;
          const float* g1 = (p.ty < remc && y < tha) ? gband + (y * a.Wo + x) * 32 : g_zero_px;
          av[jj] = f32x2{g1[0], g1[16]};
        }
      }
    };
    // operand bytes of a piece, LDS -> registers (also one piece ahead: the multiply below never waits for LDS).
    // Units past the row / the band get an address inside the band: their gradient is the zero page's.
    constexpr int NB = NT + (TAIL ? 1 : 0);
    struct Raw {
      uint32_t b[UNR][NB];
    };
    auto bload_unit = [&](const uint8_t* ub, uint32_t (&b)[NB]) {
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = ub[koff[t]];
      if (TAIL) b[NT] = ub[koff_tail];
    };
    auto bload = [&](const Piece& p, Raw& R) {
      if (p.q < nrow) {
        const uint8_t* rowb = ldsb + __mul24(2 * p.ty, RSb) + 2 * (4 * UNR * p.c + kq) * C;
        if (UNR * (p.c + 1) <= fullu) {  // the usual piece: compile-time offsets
#pragma unroll
          for (int jj = 0; jj < UNR; ++jj) bload_unit(rowb + 8 * C * jj, R.b[jj]);
        } else {
#pragma unroll
          for (int jj = 0; jj < UNR; ++jj) bload_unit(UNR * p.c + jj < fullu ? rowb + 8 * C * jj : rowb, R.b[jj]);
        }
      } else {
        const uint8_t* colb = ldsb + 2 * min(4 * fullu + p.ty, a.Wo - 1) * C;
#pragma unroll
        for (int jj = 0; jj < UNR; ++jj)
          bload_unit(colb + __mul24(2 * min(4 * (UNR * p.c + jj) + kq, tha - 1), RSb), R.b[jj]);
      }
    };
    auto mma = [&](const Raw& R, const f32x2 (&av)[UNR]) {
#pragma unroll
      for (int jj = 0; jj < UNR; ++jj) {
        float bv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = (float)R.b[jj][t];
        bsum += av[jj];
        if (TAIL) {
          const float bt = (float)R.b[jj][NB - 1];
          atail += av[jj] * f32x2{bt, bt};
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[0][t] = mfma16(av[jj][0], bv[t], acc[0][t]);
          acc[1][t] = mfma16(av[jj][1], bv[t], acc[1][t]);
        }
#ifndef CURLA_WG1_NOGROUP
        // a unit's conversions ahead of its MFMAs (a conversion right in front of the MFMA that reads it costs wait states)
        __builtin_amdgcn_sched_group_barrier(0x002, NB + 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT, 0);
#endif
      }
    };
    f32x2 avA[UNR], avB[UNR];
    Raw RA, RB;
    Piece cur;
    cur.q = ABL(2) ? npieces : wave;
    if (cur.q < nrow) cur.ty = cur.q / cpr, cur.c = cur.q - cur.ty * cpr;
    else column_of(cur);
    gload(cur, avA), bload(cur, RA);
    while (cur.q < npieces) {  // (wave-uniform)
      Piece nxt = cur;
      next_piece(nxt);
      gload(nxt, avB), bload(nxt, RB);  // (past the last piece: a column piece outside the band -- zero page, clamped bytes)
      __builtin_amdgcn_sched_barrier(0);
      mma(RA, avA);
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
      if (cur.q >= npieces) break;
      next_piece(nxt);
      gload(nxt, avA), bload(nxt, RA);
      __builtin_amdgcn_sched_barrier(0);
      mma(RB, avB);
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] *= a.scale;

  bsum[0] += __shfl_xor(bsum[0], 16);
  bsum[0] += __shfl_xor(bsum[0], 32);
  bsum[1] += __shfl_xor(bsum[1], 16);
  bsum[1] += __shfl_xor(bsum[1], 32);
  if (TAIL) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      atail[i] *= a.scale;
      atail[i] += __shfl_xor(atail[i], 16);
      atail[i] += __shfl_xor(atail[i], 32);
    }
  }
  // Cross-wave sum (waves in order: fixed order, reproducible) and the slab.  Every wave deposits its accumulator
  // tiles lane-contiguously (one ds_write_b128 per tile, no index arithmetic), as many tiles per pass as the LDS
  // holds for NW waves; after one barrier the threads add the NW copies of their slot and scatter the four sums
  // straight into the slab -- instead of NW serialised read-modify-write rounds over the output layout.
  const int nw = 32 * C * 9;
  float* slab = a.partial + (size_t)blockIdx.x * (nw + 32);
  if (ABL(4)) {  // timing only: no cross-wave sum, no slab
    float t = bsum[0] + bsum[1] + atail[0] + atail[1];
#pragma unroll
    for (int q = 0; q < 2 * NT; ++q) t += acc[q / NT][q % NT][0] + acc[q / NT][q % NT][3];
    if (t == 12345.678f) slab[tid] = t;
    return;
  }
  f32x4* l4 = reinterpret_cast<f32x4*>(lds);
  const int TC = max(1, min(2 * NT, (int)(a.lds_bytes / (NW * 1024))));  // tiles per pass (1 KB per tile and wave)
  __syncthreads();
  for (int t0 = 0; t0 < 2 * NT; t0 += TC) {
    const int nt = min(TC, 2 * NT - t0);
#pragma unroll
    for (int q = 0; q < 2 * NT; ++q)  // (tile q = mt * NT + t; compile-time register index, runtime range test)
      if (q >= t0 && q < t0 + nt) l4[(wave * TC + (q - t0)) * 64 + lane] = acc[q / NT][q % NT];
    __syncthreads();
    for (int sl = tid; sl < nt * 64; sl += NTHR) {
      const int q = t0 + sl / 64, ln = sl & 63;
      f32x4 v = l4[(0 * TC + (q - t0)) * 64 + ln];
      for (int w = 1; w < NW; ++w) v += l4[(w * TC + (q - t0)) * 64 + ln];
      const int mt = q / NT, t = q - mt * NT;
      const int k = t * 16 + (ln & 15);
      const int dy = k / (3 * C), rr = k - dy * (3 * C);
      if (dy < 3) {
        const int dx = rr / C, c = rr - dx * C;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = mt * 16 + 4 * (ln >> 4) + r;
          slab[(co * C + c) * 9 + dy * 3 + dx] = v[r];
        }
      }
    }
    __syncthreads();
  }
  if (kq == 0) {
    lds[(wave * 2 + 0) * 16 + li] = bsum[0], lds[(wave * 2 + 1) * 16 + li] = bsum[1];
    if (TAIL) lds[NW * 32 + (wave * 2 + 0) * 16 + li] = atail[0], lds[NW * 32 + (wave * 2 + 1) * 16 + li] = atail[1];
  }
  __syncthreads();
  if (tid < 32) {
    float v = lds[tid];  // wave 0: [mt][li] = tid
    for (int w = 1; w < NW; ++w) v += lds[w * 32 + tid];
    slab[nw + tid] = v;
    if (TAIL) {  // the column the tiles leave out: (dy, dx, c) = (2, 2, C-1) of output channel tid
      float t = lds[NW * 32 + tid];
      for (int w = 1; w < NW; ++w) t += lds[NW * 32 + w * 32 + tid];
      slab[(tid * C + (C - 1)) * 9 + 8] = t;
    }
  }
}


// The same weight gradient on the BF16 matrix cores (round 6; option wgrad1_u8 = auto / b16).  A uint8 pixel is EXACT in
// one bf16 (8 significand bits), the gradient is the exact sum of three (conv_rwb.h: split8) -- so a float32 product
// g x is three exact bf16 x bf16 products accumulated in fp32, nothing dropped.  One v_mfma_f32_16x16x32_bf16 takes a
// k-step of 32 PIXELS (lane group kq: pixels 8 kq .. 8 kq + 7) where the f32-input instruction takes 4: per 32 pixels
// 2 x NT x 3 matrix instructions of 16 cycles instead of 8 x 2 x NT of 32 -- 5.3 x fewer matrix cycles, which moves the
// loop from the matrix pipe (busy 0.53, 1.1 VALU per instruction) to the vector ALU: per unit and wave ~90 instructions
// split the 16 gradient values, ~70 turn the 8 x NT operand bytes into bf16 (v_cvt_f32_ubyte + one v_perm per pair: the
// float of an integer below 256 has an empty low half), ~60 walk the pixels.
// Columns: k' = (dy, rr = dx C + c) with every tap row dy padded to NTD = ceil(3 C / 16) tiles of 16 (96 columns for
// C = 9, where the unpadded 81 need six tiles as well): tile t = dy NTD + h reads byte (pixel base) + dy RSb + 16 h + li,
// i.e. ONE per-lane base per (pixel, dy) and compile-time offsets -- the padding columns multiply bytes of the
// neighbouring pixel (finite) and are dropped by the epilogue.
// Pixels: the band's pixels in row-major order, 32 per unit, units dealt round-robin to the waves (the gradient of a
// unit is 32 x 128 contiguous bytes); pixels past the band read zeros through the buffer range check and a clamped
// (valid) byte address.  What the loop waits for is the gradient (90 MB per 512 crops against 27 MB of bytes: timing-only
// builds without the byte reads, the split AND the products still take half the loop's time), so: four waves per SIMD
// (two 512-thread workgroups per CU, 128 registers) rather than three with deeper software pipelining (38 against 32 us),
// whole 128-byte lines per load instruction, and a tap row's bytes requested one tap row ahead.
template <int C, int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void wgrad1_u8b_kernel(Wgrad1Args a) {
  constexpr int NTHR = 64 * NW;
  constexpr int NTD = (3 * C + 15) / 16, NT = 3 * NTD;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using rwb::u32x4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int RSb = conv1_row_bytes(a.Wc, C);
  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0, 0, 0, 0};
  f32x2 bsum = {0.f, 0.f};
  uint8_t* ldsb = reinterpret_cast<uint8_t*>(lds);
  const int nitems = a.B * a.nbands;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int band = item / a.B, b = item - band * a.B;
    const int y0 = band * a.th;
    const int tha = min(a.th, a.Ho - y0);
    const int npix = tha * a.Wo;
    const int nunits = (npix + 31) >> 5;
    // the band's gradient, [pixel][32]: lane (li, kq) takes output channels 2 li and 2 li + 1 (rows li of the two channel
    // tiles: tile mt holds the channels of parity mt) of its 8 pixels -- ONE 8-byte load per pixel, a lane group reads a
    // pixel's whole 128-byte line (channel li and 16 + li as two 4-byte loads: twice the load instructions, each touching
    // half a line -- the loop waits for these loads, not for arithmetic); past the band: zeros
    const __amdgpu_buffer_rsrc_t rg = rw::uniform_rsrc(a.g + ((size_t)(b * a.Ho + y0) * a.Wo) * 32, npix * 128);
    float graw[2][8];
    auto gload = [&](int u) {
      const unsigned v0 = (unsigned)((32 * u + 8 * kq) * 128 + li * 8);
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rg, v0 + 128u * jj, 0, 0));
        graw[0][jj] = v[0], graw[1][jj] = v[1];
      }
    };
    if (wave < nunits) gload(wave);  // (independent of the crop: in flight while the bytes are staged)
    if (!ABL(1)) {
      const int64_t fi = a.idx ? rw::const_load(a.idx, b) : (int64_t)b;  // (scalar loads: b is wave-uniform)
      const int oh = a.h1 ? rw::const_load(a.h1, b) : 0, ow = a.w1 ? rw::const_load(a.w1, b) : 0;
      const uint8_t* crop = static_cast<const uint8_t*>(a.src) + ((size_t)fi * a.Hs + oh + 2 * y0) * a.Ws * C + (size_t)ow * C;
      if (a.Wc * C <= 64 * 16)  // a lane per 16-byte run of a row
        conv1_stage_rows<10>(ldsb, crop, a.Ws * C, a.Wc * C, 2 * tha + 1, RSb, 0, wave, NW, lane);
      else
        conv1_stage_u8(ldsb, static_cast<const uint8_t*>(a.src), a.idx, a.h1, a.w1, b, C, a.Hs, a.Ws, a.Wc, 2 * y0,
                       2 * tha + 1, RSb, tid, NTHR);
    }
    __syncthreads();
    // first pixel of this lane in unit u = wave: (row, column) -> byte offset of its patch; a unit step is 32 NW pixels
    int p0 = 32 * wave + 8 * kq;
    int ty = p0 / a.Wo, x = p0 - ty * a.Wo;
    const int qstep = (32 * NW) / a.Wo, rstep = 32 * NW - qstep * a.Wo;
    const int px = 2 * C, wrap_add = 2 * RSb - 2 * (a.Wo - 1) * C;  // next pixel of a row / first pixel of the next row
    const int pb_last = 2 * (tha - 1) * RSb + 2 * (a.Wo - 1) * C + li;  // (pixels past the band: clamped, zero gradient)
    for (int u = ABL(2) ? nunits : wave; u < nunits; u += NW) {  // (wave-uniform)
      // ---- byte address of each of the 8 pixels' patch and the first tap row's bytes, in flight during the split
      int pb[8];
      {
        int xx = x, cur = __mul24(2 * ty, RSb) + 2 * x * C + li;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          pb[jj] = min(cur, pb_last);
          ++xx;
          const bool wrap = xx >= a.Wo;
          cur += wrap ? wrap_add : px;
          xx = wrap ? 0 : xx;
        }
      }
      uint32_t raw[8 * NTD];
      auto bread = [&](const int dy) {
        const uint8_t* rowb = ldsb + dy * RSb;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
#pragma unroll
          for (int h = 0; h < NTD; ++h) raw[jj * NTD + h] = rowb[pb[jj] + 16 * h];
      };
      if (!ABL(8)) bread(0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- this unit's gradient values -> three bf16 operands per channel half; the next unit's are requested
      rwb::B3 G[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = graw[mt][jj];
        bsum[mt] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        if (ABL(16)) {  // timing only: no split arithmetic
#pragma unroll
          for (int q = 0; q < 4; ++q)
            G[mt].h[q] = __builtin_bit_cast(unsigned, v[q]), G[mt].m[q] = __builtin_bit_cast(unsigned, v[q + 4]), G[mt].l[q] = G[mt].h[q];
        } else
          G[mt] = rwb::split8(v);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- tap row by tap row: 8 NTD bytes -> NTD B operands of 8 bf16 (the float of an integer below 256 has an empty
      // low half: its high half IS the bf16), the next tap row's bytes requested, then the six products of each tile
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        u32x4 X[NTD];
#pragma unroll
        for (int h = 0; h < NTD; ++h)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float f0 = (float)raw[(2 * q) * NTD + h], f1 = (float)raw[(2 * q + 1) * NTD + h];
            X[h][q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, f1), __builtin_bit_cast(unsigned, f0), 0x07060302u);
          }
        __builtin_amdgcn_sched_barrier(0);
        if (dy < 2 && !ABL(8)) bread(dy + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (ABL(32)) {  // timing only: no matrix instructions
#pragma unroll
          for (int h = 0; h < NTD; ++h) acc[0][dy * NTD + h][0] += __builtin_bit_cast(float, X[h][0] ^ X[h][1] ^ X[h][2] ^ X[h][3] ^ G[0].l[0] ^ G[1].m[1] ^ G[0].h[2] ^ G[1].h[3] ^ G[0].m[0] ^ G[1].l[1]);
        } else
#pragma unroll
        for (int h = 0; h < NTD; ++h) {
          const int t = dy * NTD + h;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt][t] = rwb::mfma_bf16(G[mt].l, X[h], acc[mt][t]);  // smallest part first
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt][t] = rwb::mfma_bf16(G[mt].m, X[h], acc[mt][t]);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt][t] = rwb::mfma_bf16(G[mt].h, X[h], acc[mt][t]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // the next unit's gradient values: requested here, behind the unit's arithmetic (their 16 registers are free
      // again; the SIMD's other three waves cover the latency -- requested before the products they cost 33 spills)
      if (u + NW < nunits) gload(u + NW);
      x += rstep, ty += qstep;  // at most one more wrap
      const bool wrap = x >= a.Wo;
      x = wrap ? x - a.Wo : x;
      ty = wrap ? ty + 1 : ty;
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] *= a.scale;
  bsum[0] += __shfl_xor(bsum[0], 16);
  bsum[0] += __shfl_xor(bsum[0], 32);
  bsum[1] += __shfl_xor(bsum[1], 16);
  bsum[1] += __shfl_xor(bsum[1], 32);
  // cross-wave sum in wave order and the slab, as wgrad1_u8_kernel (tile t: tap row t / NTD, columns 16 (t % NTD) + li)
  const int nw = 32 * C * 9;
  float* slab = a.partial + (size_t)blockIdx.x * (nw + 32);
  if (ABL(4)) {  // timing only: no cross-wave sum, no slab
    float t = bsum[0] + bsum[1];
#pragma unroll
    for (int q = 0; q < 2 * NT; ++q) t += acc[q / NT][q % NT][0] + acc[q / NT][q % NT][3];
    if (t == 12345.678f) slab[tid] = t;
    return;
  }
  f32x4* l4 = reinterpret_cast<f32x4*>(lds);
  const int TC = max(1, min(2 * NT, (int)(a.lds_bytes / (NW * 1024))));  // tiles per pass (1 KB per tile and wave)
  __syncthreads();
  for (int t0 = 0; t0 < 2 * NT; t0 += TC) {
    const int nt = min(TC, 2 * NT - t0);
#pragma unroll
    for (int q = 0; q < 2 * NT; ++q)
      if (q >= t0 && q < t0 + nt) l4[(wave * TC + (q - t0)) * 64 + lane] = acc[q / NT][q % NT];
    __syncthreads();
    for (int sl = tid; sl < nt * 64; sl += NTHR) {
      const int q = t0 + sl / 64, ln = sl & 63;
      f32x4 v = l4[(0 * TC + (q - t0)) * 64 + ln];
      for (int w = 1; w < NW; ++w) v += l4[(w * TC + (q - t0)) * 64 + ln];
      const int mt = q / NT, t = q - mt * NT;
      const int dy = t / NTD, rr = 16 * (t - dy * NTD) + (ln & 15);
      if (rr < 3 * C) {
        const int dx = rr / C, c = rr - dx * C;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = 2 * (4 * (ln >> 4) + r) + mt;  // (tile mt: the output channels of parity mt)
          slab[(co * C + c) * 9 + dy * 3 + dx] = v[r];
        }
      }
    }
    __syncthreads();
  }
  if (kq == 0) lds[(wave * 2 + 0) * 16 + li] = bsum[0], lds[(wave * 2 + 1) * 16 + li] = bsum[1];
  __syncthreads();
  if (tid < 32) {  // tid = mt * 16 + li: output channel 2 li + mt
    float v = lds[tid];
    for (int w = 1; w < NW; ++w) v += lds[w * 32 + tid];
    slab[nw + 2 * (tid & 15) + (tid >> 4)] = v;
  }
}


// second pass: dW = sum over workgroup slabs.  32 elements x 32 slab-groups per
// block; each group adds its slabs in slab order, the 32 group sums are added in
// group order (fixed order => bitwise reproducible).
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* partial, int nslabs, int nw, int nb, float* dw,
                                                            float* db) {
  __shared__ float sm[32][33];
  const int c = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + c;
  const int n = nw + nb;  // (nb bias sums behind the nw weight sums of a slab: 32, or the filter count of the generic path)
  float s = 0.f;
  if (i < n) {
    int k = part;
    for (; k + 7 * 32 < nslabs; k += 8 * 32) {  // 8 slabs in flight, added in slab order
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = partial[(size_t)(k + 32 * u) * n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; k < nslabs; k += 32) s += partial[(size_t)k * n + i];
  }
  sm[part][c] = s;
  __syncthreads();
  if (part == 0 && i < n) {
    float t = sm[0][c];
#pragma unroll
    for (int k = 1; k < 32; ++k) t += sm[k][c];
    if (i < nw)
      dw[i] = t;
    else
      db[i - nw] = t;
  }
}

// the same for up to kMaxReduceJobs weight gradients in ONE launch: the backward pass of an encoder leaves one slab set
// per conv layer and nothing reads dW before the pass is over, so the per-layer reductions need not be launches of
// their own (each would cost the 4.8 us launch floor for ~1 us of work)
constexpr int kMaxReduceJobs = 8;
struct ReduceJobs {
  const float* partial[kMaxReduceJobs];
  float* dw[kMaxReduceJobs];
  float* db[kMaxReduceJobs];
  int nslabs[kMaxReduceJobs], nw[kMaxReduceJobs], nb[kMaxReduceJobs], first_block[kMaxReduceJobs + 1];
  int njobs;
};

__global__ __launch_bounds__(1024) void wgrad_reduce_multi_kernel(ReduceJobs J) {
  __shared__ float sm[32][33];
  int j = 0;
  while (j + 1 < J.njobs && (int)blockIdx.x >= J.first_block[j + 1]) ++j;
  const float* partial = J.partial[j];
  const int nslabs = J.nslabs[j], nw = J.nw[j];
  const int c = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int i = ((int)blockIdx.x - J.first_block[j]) * 32 + c;
  const int n = nw + J.nb[j];
  float s = 0.f;
  if (i < n) {
    int k = part;
    for (; k + 7 * 32 < nslabs; k += 8 * 32) {  // 8 slabs in flight, added in slab order
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = partial[(size_t)(k + 32 * u) * n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; k < nslabs; k += 32) s += partial[(size_t)k * n + i];
  }
  sm[part][c] = s;
  __syncthreads();
  if (part == 0 && i < n) {
    float t = sm[0][c];
#pragma unroll
    for (int k = 1; k < 32; ++k) t += sm[k][c];
    if (i < nw)
      J.dw[j][i] = t;
    else
      J.db[j][i - nw] = t;
  }
}

// ------------------------------ host-side planning ------------------------------
constexpr int kMaxLds = 160 * 1024;
static_assert(2 * rwb::kWBytes <= kMaxLds, "the two problems' split filters of conv_rwb_fwd_kernel must fit one CU's LDS");
// Dynamic LDS limit of a kernel: raised once per (kernel, device) to the largest size this library ever asks for --
// never lowered, never set per launch (curla_set_dyn_lds, common.h).
template <typename K>
int set_lds(K kernel, size_t bytes) {
  if (bytes > (size_t)kMaxLds) return CURLA_ERR_UNSUPPORTED;
  return curla_set_dyn_lds(reinterpret_cast<const void*>(kernel), kMaxLds);
}

// Which uint8 first-layer forward runs (option conv1_u8, options.h): "rw" = the LDS-free row walk (conv1_u8_rw.h),
// "hybrid" = conv1_u8_walk_kernel (crop staged in LDS as bytes, row walk out of LDS) whenever the crop fits one band of
// LDS, else the banded loop, "band" = the banded loop.  Measured on 1024 + 512 / 512 + 512 samples of configs[1]: alone,
// re-reading the same ring slots out of the Infinity Cache, the LDS-free walk takes 100 / 66 us against the banded
// loop's 128 / 86; on slots drawn afresh for every launch from a ring of gigabytes -- what update() does -- it was the
// slower one in rounds 3-4 (114 us on average against the hybrid's 102) and is the faster one since the stride-1
// convs around it run on the bf16 matrix cores (round 5, whole update, alternating runs on one box: configs[1] 493.4 /
// 493.4 against 489.0 / 490.2 update()/s, configs[2] 634.3 against 632.0).  auto = rw where it applies.
bool use_rw_u8() { return curla_opt(kOptConv1U8) == 3 || curla_opt(kOptConv1U8) == 0 || curla_opt(kOptConv1U8) == 4; }

// The row-walk forward keeps (pixel pair, 32 channels) of a whole row in flight per wave; any width works, the strips
// only get more numerous.  Limits: byte offsets inside one sample must fit 31 bits.
bool rw_supported(int Hi, int Wi) { return (long long)(Hi + 2) * (Wi + 2) * 128 < (1LL << 30); }

int launch_rw_fwd(int nlayers, const float* in, const float* const* w, const float* const* bias, float* const* out, int B,
                  const float* in2, const float* const* w2, const float* const* bias2, float* const* out2, int B2, int Hi,
                  int Wi, bool owned, hipStream_t st) {
  // Three forms (option s1_fwd).  auto = b3, the bf16-matrix-core form (conv_rwb.h: fp32 operands as three bf16 parts,
  // Winograd F(2,3) in front): measured on the stacks update() launches (tools/s1_bench.py) it takes 365 / 253 us on
  // configs[1]'s two stacks against 491 / 337 for F(2,3) and 451 / 309 for F(4,3) on the f32-input MFMA, and 5.99 / 4.01 ms
  // against F(4,3)'s 7.15 / 4.71 on configs[4]'s.  f23 / f43 (conv_rw.h / conv_rw43.h) remain selectable: F(4,3) issues
  // 0.78 x the f32 MFMAs of F(2,3) but ~0.9 VALU instructions per MFMA instead of 0.4 at one wave per SIMD, and wins among
  // the two where rows hold at least one full strip of 16 pixel quads.
  const int opt43 = curla_opt(kOptS1Fwd);
  const bool b3 = opt43 == 3 || opt43 == 0;
  const bool f43 = opt43 == 2;
  rw::Args A;
  A.nlayers = nlayers;
  for (int l = 0; l < rw::kMaxLayers; ++l) {
    const bool on = l < nlayers;
    const int hi = Hi - 2 * l, wi = Wi - 2 * l;
    A.g[l] = !on ? rw::Geom{}
             : b3 ? rwb::plan(hi, wi, hi - 2, wi - 2)
                  : (f43 ? rw43::plan(hi, wi, hi - 2, wi - 2) : rw::plan(hi, wi, hi - 2, wi - 2));
    A.p[l][0] = on ? rw::Problem{l == 0 ? in : out[l - 1], w[l], bias[l], out[l], B} : rw::Problem{};
    A.p[l][1] = (on && B2 > 0) ? rw::Problem{l == 0 ? in2 : out2[l - 1], w2[l], bias2[l], out2[l], B2} : rw::Problem{};
    if (on && (hi < 3 || wi < 3)) return CURLA_ERR_UNSUPPORTED;
  }
  if (!rw_supported(Hi, Wi)) return CURLA_ERR_UNSUPPORTED;
  const int cus = curla_cu_count();
  const int bmax = B > B2 ? B : B2;
  const int grid = owned ? cus : (bmax < cus ? bmax : cus);
  if (b3) {
    const size_t ldsb = (size_t)(B2 > 0 ? 2 : 1) * rwb::kWBytes;
    int rcb = set_lds(conv_rwb_fwd_kernel, ldsb);
    if (rcb != CURLA_OK) return rcb;
    hipLaunchKernelGGL(conv_rwb_fwd_kernel, dim3(grid), dim3(512), ldsb, st, A);
    return curla_launch_status();
  }
  if (f43) {
    const size_t lds43 = (size_t)(B2 > 0 ? 2 : 1) * rw43::kWFloats * sizeof(float);
    int rc43 = set_lds(conv_rw43_fwd_kernel, lds43);
    if (rc43 != CURLA_OK) return rc43;
    hipLaunchKernelGGL(conv_rw43_fwd_kernel, dim3(grid), dim3(256), lds43, st, A);
    return curla_launch_status();
  }
  const size_t lds = (size_t)(B2 > 0 ? 2 : 1) * rw::kWFloats * sizeof(float);
  int rc = set_lds(conv_rw_fwd_kernel, lds);
  if (rc != CURLA_OK) return rc;
  hipLaunchKernelGGL(conv_rw_fwd_kernel, dim3(grid), dim3(512), lds, st, A);
  return curla_launch_status();
}

// data-gradient arguments: input = the layer's output gradient [B][Ho][Wo][32], output [B][Ho+2][Wo+2][32]
rw::Args rw_dgrad_args(const float* g, const float* w, const float* act_below, float* gin, int B, int Ho, int Wo) {
  rw::Args A;
  A.nlayers = 1;
  for (int l = 0; l < rw::kMaxLayers; ++l) A.g[l] = rw::Geom{}, A.p[l][0] = rw::Problem{}, A.p[l][1] = rw::Problem{};
  A.g[0] = rw::plan(Ho, Wo, Ho + 2, Wo + 2);
  A.p[0][0] = rw::Problem{g, w, act_below, gin, B};
  return A;
}

// Does the data gradient of a layer whose INPUT is Wi wide take the F(4,3) kernel?  (option s1_fwd, as the forward)
bool dgrad_f43(int Wi) {
  const int opt = curla_opt(kOptS1Fwd);
  return opt == 2 || (opt == 0 && (Wi + 3) / 4 >= 16);
}

// ... or the bf16x3 kernel (conv_rwb.h)?  (auto: yes -- beside the weight gradient's workgroups in one launch it takes
// configs[1] from 474.6 to 487.3 update()/s and configs[4] from 34.6 to 35.2 against the F(2,3) / F(4,3) forms)
bool dgrad_b3() { return curla_opt(kOptS1Fwd) == 3 || curla_opt(kOptS1Fwd) == 0; }

int launch_dgrad_b3(const float* g, const float* w, const float* act_below, float* gin, int B, int Ho, int Wo,
                    hipStream_t st) {
  rw::Args A = rw_dgrad_args(g, w, act_below, gin, B, Ho, Wo);
  A.g[0] = rwb::plan(Ho, Wo, Ho + 2, Wo + 2);
  const int cap = curla_cu_count();
  int rc = set_lds(conv_rwb_dgrad_kernel, rwb::kWBytes);
  if (rc != CURLA_OK) return rc;
  hipLaunchKernelGGL(conv_rwb_dgrad_kernel, dim3(B < cap ? B : cap), dim3(512), rwb::kWBytes, st, A);
  return curla_launch_status();
}

int launch_dgrad43(const float* g, const float* w, const float* act_below, float* gin, int B, int Ho, int Wo,
                   hipStream_t st) {
  rw::Args A = rw_dgrad_args(g, w, act_below, gin, B, Ho, Wo);
  A.g[0] = rw43::plan(Ho, Wo, Ho + 2, Wo + 2);
  const int cap = curla_cu_count();
  const size_t lds = rw43::kWFloats * sizeof(float);
  int rc = set_lds(conv_rw43_dgrad_kernel, lds);
  if (rc != CURLA_OK) return rc;
  hipLaunchKernelGGL(conv_rw43_dgrad_kernel, dim3(B < cap ? B : cap), dim3(256), lds, st, A);
  return curla_launch_status();
}

int launch_conv_s1(int mode, const float* in, const float* w, const float* aux, float* out, int B, int Hs, int Ws,
                   hipStream_t st, const float* in2 = nullptr, const float* w2 = nullptr, const float* aux2 = nullptr,
                   float* out2 = nullptr, int B2 = 0) {
  if (!rw_supported(Hs, Ws)) return CURLA_ERR_UNSUPPORTED;
  if (mode == MODE_FWD) return launch_rw_fwd(1, in, &w, &aux, &out, B, in2, &w2, &aux2, &out2, B2, Hs, Ws, false, st);
  if (dgrad_b3()) return launch_dgrad_b3(in, w, aux, out, B, Hs, Ws, st);
  if (dgrad_f43(Ws + 2)) return launch_dgrad43(in, w, aux, out, B, Hs, Ws, st);
  const rw::Args A = rw_dgrad_args(in, w, aux, out, B, Hs, Ws);
  const int cap = 2 * curla_cu_count();
  const size_t lds = rw::kWFloats * sizeof(float);
  int rc = set_lds(conv_rw_dgrad_kernel, lds);
  if (rc != CURLA_OK) return rc;
  hipLaunchKernelGGL(conv_rw_dgrad_kernel, dim3(B < cap ? B : cap), dim3(256), lds, st, A);
  return curla_launch_status();
}

int plan_band_conv1(int Ho, int Wo, int Wc, int C, int g_px_per_row, size_t lds_budget) {
  const int RS = ((Wc * C + 3) & ~3) + 4;
  int best = 1;
  double best_eff = -1.0;
  for (int th = 1; th <= Ho; ++th) {
    const size_t bytes = ((size_t)(2 * th + 1) * RS + (size_t)g_px_per_row * th * kLdsPix) * sizeof(float);
    if (bytes > lds_budget) break;
    const int nb = (Ho + th - 1) / th;
    double work = 0, slots = 0;
    for (int bnd = 0; bnd < nb; ++bnd) {
      const int tha = (bnd == nb - 1) ? Ho - bnd * th : th;
      const int tiles = (tha * Wo + 15) / 16;
      work += tha * Wo / 16.0;
      slots += ((tiles + 7) / 8) * 8;
    }
    const double eff = work / slots * (2.0 * th) / (2.0 * th + 1 + 2);
    if (eff > best_eff) best_eff = eff, best = th;
  }
  return best;
}

}  // namespace

// ------------------------------------ C ABI ------------------------------------
extern "C" {

#ifdef CURLA_ABLATE
void curla_debug_ablate(int flags) { g_ablate = flags; }
#endif
#ifdef RWB_CLOCK
int curla_debug_rwb_clock(unsigned long long* out, int n_workgroups, int reset) {
  if (n_workgroups < 1 || n_workgroups > 1024) return -1;
  if (reset) {
    static unsigned long long z[3 * 1024];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_rwb_clock), z, sizeof(z)) == hipSuccess ? 0 : -2;
  }
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rwb_clock), 3 * n_workgroups * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif
#ifdef RWB_STAMP
// timing-only debug build (tools/build_variant.sh stamp -DRWB_STAMP): cycle sums of wave 0 of workgroup 7 over its full steps
int curla_debug_rwb_stamps(unsigned long long* out8, int reset) {
  if (reset) {
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(rwb::g_rwb_stamp), z, sizeof(z)) == hipSuccess ? 0 : -2;
  }
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(rwb::g_rwb_stamp), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

}  // extern "C"  (re-opened below: the generic-width helpers are C++)
#include "conv_generic.h"

namespace {
// ---- filter counts other than 32: the plain kernels of conv_generic.h behind the same entry points -------------------
int gen_fwd_s1(const float* in, const float* w, const float* bias, float* out, int B, int Hi, int Wi, int C, hipStream_t st) {
  if (!gen::channels_ok(C)) return CURLA_ERR_UNSUPPORTED;
  const int Ho = Hi - 2, Wo = Wi - 2;
  gen::Src none{};
  hipLaunchKernelGGL(gen::conv_fwd_kernel<false>, dim3(gen::grid_for((size_t)B * Ho * Wo * C)), dim3(256), 0, st, none, in, w,
                     bias, out, B, Hi, Wi, C, Ho, Wo, C);
  return curla_launch_status();
}

gen::Src gen_src(const void* src, int src_kind, const int64_t* idx, const int32_t* h1, const int32_t* w1, int C, int Hs, int Ws,
                 int Hc, int Wc, float scale) {
  return gen::Src{src, src_kind, idx, h1, w1, C, Hs, Ws, Hc, Wc, scale};
}

int gen_fwd1(const gen::Src& s, const float* w, const float* bias, float* out, int B, int channels, hipStream_t st) {
  if (!gen::channels_ok(channels)) return CURLA_ERR_UNSUPPORTED;
  const int Ho = (s.Hc - 3) / 2 + 1, Wo = (s.Wc - 3) / 2 + 1;
  hipLaunchKernelGGL(gen::conv_fwd_kernel<true>, dim3(gen::grid_for((size_t)B * Ho * Wo * channels)), dim3(256), 0, st, s,
                     static_cast<const float*>(nullptr), w, bias, out, B, s.Hc, s.Wc, s.C, Ho, Wo, channels);
  return curla_launch_status();
}

int gen_dgrad(const float* g, const float* w, const float* act_below, float* gin, int B, int Ho, int Wo, int C, hipStream_t st) {
  if (!gen::channels_ok(C)) return CURLA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gen::conv_dgrad_kernel, dim3(gen::grid_for((size_t)B * (Ho + 2) * (Wo + 2) * C)), dim3(256), 0, st, g, w,
                     act_below, gin, B, Ho, Wo, C);
  return curla_launch_status();
}

// (one slab: [channels * cin * 9 | channels])
int gen_wgrad_s1(const float* in, const float* g, float* slab, int B, int Hi, int Wi, int C, hipStream_t st, int* nslabs) {
  if (!gen::channels_ok(C)) return CURLA_ERR_UNSUPPORTED;
  gen::Src none{};
  hipLaunchKernelGGL(gen::conv_wgrad_kernel<false>, dim3(C * C + C), dim3(256), 0, st, none, in, g, slab, B, Hi, Wi, C, Hi - 2,
                     Wi - 2, C);
  *nslabs = 1;
  return curla_launch_status();
}

int gen_wgrad1(const gen::Src& s, const float* g, float* slab, int B, int channels, hipStream_t st, int* nslabs) {
  if (!gen::channels_ok(channels)) return CURLA_ERR_UNSUPPORTED;
  const int Ho = (s.Hc - 3) / 2 + 1, Wo = (s.Wc - 3) / 2 + 1;
  hipLaunchKernelGGL(gen::conv_wgrad_kernel<true>, dim3(channels * s.C + channels), dim3(256), 0, st, s,
                     static_cast<const float*>(nullptr), g, slab, B, s.Hc, s.Wc, s.C, Ho, Wo, channels);
  *nslabs = 1;
  return curla_launch_status();
}
}  // namespace

extern "C" {

int curla_conv3x3_s1_fwd(const float* in, const float* w, const float* bias, float* out, int B, int Hi, int Wi,
                         int channels, void* stream) {
  CURLA_REQUIRE(in && w && bias && out && B > 0 && Hi >= 3 && Wi >= 3);
  if (channels != 32) return gen_fwd_s1(in, w, bias, out, B, Hi, Wi, channels, static_cast<hipStream_t>(stream));
  CURLA_REQUIRE(aligned16(in) && aligned16(out) && aligned16(bias) && aligned16(w));
  return launch_conv_s1(MODE_FWD, in, w, bias, out, B, Hi, Wi, static_cast<hipStream_t>(stream));
}

int curla_conv3x3_s1_fwd2(const float* in, const float* w, const float* bias, float* out, int B, const float* in2,
                          const float* w2, const float* bias2, float* out2, int B2, int Hi, int Wi, int channels,
                          void* stream) {
  CURLA_REQUIRE(in && w && bias && out && B > 0 && in2 && w2 && bias2 && out2 && B2 > 0 && Hi >= 3 && Wi >= 3);
  if (channels != 32) {  // (generic width: the two problems one after the other)
    const int rc = gen_fwd_s1(in, w, bias, out, B, Hi, Wi, channels, static_cast<hipStream_t>(stream));
    return rc != CURLA_OK ? rc : gen_fwd_s1(in2, w2, bias2, out2, B2, Hi, Wi, channels, static_cast<hipStream_t>(stream));
  }
  CURLA_REQUIRE(aligned16(in) && aligned16(out) && aligned16(bias) && aligned16(w));
  CURLA_REQUIRE(aligned16(in2) && aligned16(out2) && aligned16(bias2) && aligned16(w2));
  return launch_conv_s1(MODE_FWD, in, w, bias, out, B, Hi, Wi, static_cast<hipStream_t>(stream), in2, w2, bias2, out2, B2);
}

int curla_conv3x3_s1_fwd_stack(int nlayers, const float* in, const float* const* w, const float* const* bias,
                               float* const* out, int B, const float* in2, const float* const* w2,
                               const float* const* bias2, float* const* out2, int B2, int Hi, int Wi, int channels,
                               void* stream) {
  CURLA_REQUIRE(nlayers > 0 && nlayers <= rw::kMaxLayers && in && w && bias && out && B > 0 && Hi >= 3 && Wi >= 3);
  CURLA_REQUIRE(B2 == 0 || (in2 && w2 && bias2 && out2));
  if (channels != 32) return CURLA_ERR_UNSUPPORTED;
  // ownership of samples by workgroups needs whole rounds of the grid (one workgroup per CU) over each minibatch
  const int G1 = curla_cu_count();
  if (B % G1 != 0 || B2 % G1 != 0) return CURLA_ERR_UNSUPPORTED;
  CURLA_REQUIRE(aligned16(in) && (!B2 || aligned16(in2)));
  for (int l = 0; l < nlayers; ++l) {
    CURLA_REQUIRE(w[l] && bias[l] && out[l] && aligned16(w[l]) && aligned16(bias[l]) && aligned16(out[l]));
    CURLA_REQUIRE(!B2 || (w2[l] && bias2[l] && out2[l] && aligned16(w2[l]) && aligned16(bias2[l]) && aligned16(out2[l])));
  }
  if (Hi - 2 * nlayers < 1 || Wi - 2 * nlayers < 1) return CURLA_ERR_UNSUPPORTED;
  return launch_rw_fwd(nlayers, in, w, bias, out, B, in2, w2, bias2, out2, B2, Hi, Wi, true,
                       static_cast<hipStream_t>(stream));
}

int curla_conv3x3_s1_stack_granule(void) { return curla_cu_count(); }

int curla_conv3x3_s1_dgrad(const float* g, const float* w, const float* act_below, float* gin, int B, int Ho, int Wo,
                           int channels, void* stream) {
  CURLA_REQUIRE(g && w && act_below && gin && B > 0 && Ho >= 1 && Wo >= 1);
  if (channels != 32) return gen_dgrad(g, w, act_below, gin, B, Ho, Wo, channels, static_cast<hipStream_t>(stream));
  CURLA_REQUIRE(aligned16(g) && aligned16(gin) && aligned16(act_below) && aligned16(w));
  return launch_conv_s1(MODE_DGRAD, g, w, act_below, gin, B, Ho, Wo, static_cast<hipStream_t>(stream));
}

static int conv1_common_check(const void* src, int src_is_u8, int B, int C, int Hs, int Ws, int Hc, int Wc,
                              const int32_t* h1, const int32_t* w1) {
  CURLA_REQUIRE(src && B > 0 && Hc >= 3 && Wc >= 3 && src_is_u8 >= 0 && src_is_u8 <= 2);
  if (C != 9 && C != 12 && C != 6 && C != 3) return CURLA_ERR_UNSUPPORTED;  // 3 x frame_stack of 1..4
  if (src_is_u8 == 1) {
    CURLA_REQUIRE(Hs >= Hc && Ws >= Wc);
    // the loader rebuilds every 16-byte run from aligned dwords whatever its byte address, so frames of any size
    // work; only the ring's base must be dword-aligned (the first run would otherwise start before the buffer)
    CURLA_REQUIRE((reinterpret_cast<uintptr_t>(src) & 3) == 0);
    (void)h1, (void)w1;
  }
  return CURLA_OK;
}

#define CURLA_DISPATCH_SRC(CC, KIND, KERNEL, ...)                  \
  do {                                                             \
    if ((KIND) == 1) {                                             \
      KERNEL(SRC_U8, CC, __VA_ARGS__);                             \
    } else if ((KIND) == 2) {                                      \
      KERNEL(SRC_NHWC, CC, __VA_ARGS__);                           \
    } else {                                                       \
      KERNEL(SRC_F32, CC, __VA_ARGS__);                            \
    }                                                              \
  } while (0)
#define CURLA_DISPATCH_C(C, KIND, KERNEL, ...)                     \
  do {                                                             \
    if ((C) == 9) {                                                \
      CURLA_DISPATCH_SRC(9, KIND, KERNEL, __VA_ARGS__);            \
    } else if ((C) == 12) {                                        \
      CURLA_DISPATCH_SRC(12, KIND, KERNEL, __VA_ARGS__);           \
    } else if ((C) == 6) {                                         \
      CURLA_DISPATCH_SRC(6, KIND, KERNEL, __VA_ARGS__);            \
    } else {                                                       \
      CURLA_DISPATCH_SRC(3, KIND, KERNEL, __VA_ARGS__);            \
    }                                                              \
  } while (0)

#define CONV1_FWD_LAUNCH(SRC, CC, grid, lds, st, a)                                       \
  {                                                                                       \
    rc = set_lds(conv1_fwd_kernel<SRC, CC>, lds);                                         \
    if (rc == CURLA_OK) hipLaunchKernelGGL((conv1_fwd_kernel<SRC, CC>), dim3(grid), dim3(512), lds, st, a); \
  }

struct Conv1Second {
  const int64_t* idx;
  const int32_t* h1;
  const int32_t* w1;
  const float* w;
  const float* bias;
  float* out;
  int B;
};

static int conv1_fwd_impl(const void* src, int src_kind, const int64_t* idx, const int32_t* h1, const int32_t* w1,
                          const float* w, const float* bias, float* out, int B, int C, int Hs, int Ws, int Hc, int Wc,
                          int channels, float scale, void* stream, const Conv1Second* second) {
  CURLA_REQUIRE(w && bias && out);
  int rc = conv1_common_check(src, src_kind, B, C, Hs, Ws, Hc, Wc, h1, w1);
  if (rc != CURLA_OK) return rc;
  if (channels != 32) {  // generic width (conv_generic.h): the second minibatch, if any, as a launch of its own
    if (second && src_kind != 1) return CURLA_ERR_UNSUPPORTED;
    hipStream_t gst = static_cast<hipStream_t>(stream);
    rc = gen_fwd1(gen_src(src, src_kind, idx, h1, w1, C, Hs, Ws, Hc, Wc, scale), w, bias, out, B, channels, gst);
    if (rc == CURLA_OK && second)
      rc = gen_fwd1(gen_src(src, src_kind, second->idx, second->h1, second->w1, C, Hs, Ws, Hc, Wc, scale), second->w,
                    second->bias, second->out, second->B, channels, gst);
    return rc;
  }
  Conv1Args a;
  a.src = src, a.idx = idx, a.h1 = h1, a.w1 = w1, a.w = w, a.bias = bias, a.out = out;
  a.B = B, a.C = C, a.Hs = Hs, a.Ws = Ws, a.Hc = Hc, a.Wc = Wc;
  a.Ho = (Hc - 3) / 2 + 1, a.Wo = (Wc - 3) / 2 + 1;
  a.scale = scale;
  a.dbg = ABL_HOST;
  a.idx2 = second ? second->idx : nullptr, a.h1_2 = second ? second->h1 : nullptr, a.w1_2 = second ? second->w1 : nullptr;
  a.w2 = second ? second->w : nullptr, a.bias2 = second ? second->bias : nullptr, a.out2 = second ? second->out : nullptr;
  a.B2 = second ? second->B : 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t wl = (size_t)32 * C * 9 * sizeof(float);
  if (second && src_kind != 1) return CURLA_ERR_UNSUPPORTED;
  if (src_kind == 1 && use_rw_u8() && (long long)Hs * Ws * C < (1LL << 30) &&
      (long long)(B + a.B2) * a.Ho * ((a.Wo + 15) / 16 + 1) < (1LL << 28)) {
    // uint8 ring: row walk, nothing staged (conv1_u8_rw.h)
    rw::Conv1U8Args ra;
    ra.src = static_cast<const uint8_t*>(src);
    ra.p[0] = rw::Conv1U8Problem{idx, h1, w1, w, bias, out, B};
    ra.p[1] = rw::Conv1U8Problem{a.idx2, a.h1_2, a.w1_2, a.w2, a.bias2, a.out2, a.B2};
    ra.Hs = Hs, ra.Ws = Ws, ra.Ho = a.Ho, ra.Wo = a.Wo, ra.scale = scale;
    ra.g.Hi = Hc, ra.g.Wi = Wc, ra.g.Ho = a.Ho, ra.g.Wo = a.Wo;
    rw::plan_units(ra.g, a.Ho, a.Wo, 16);
    // kC1U8PerCU workgroups per CU; fewer when a workgroup's share of the pool would drop below a few steps per wave
    const int cap = kC1U8PerCU * curla_cu_count();
    const long long pool = (long long)(B + a.B2) * ra.g.steps;
    const int want = (int)((pool + 63) / 64);
    const int grid_rw = want < cap ? (want < 1 ? 1 : want) : cap;
    // the bf16 form where an input row's 3 C operand bytes are one k-step of 32 (option conv1_u8 = auto / rwb)
    const int o8 = curla_opt(kOptConv1U8);
    if ((o8 == 0 || o8 == 4) && 3 * C <= 32) {
      const int capb = kC1U8bPerCU * curla_cu_count();
      const int wantb = (int)((pool + 31) / 32);
      const int grid_rwb = wantb < capb ? (wantb < 1 ? 1 : wantb) : capb;
#define CONV1_U8_RWB_LAUNCH(CC) hipLaunchKernelGGL((conv1_u8_rwb_fwd_kernel<CC>), dim3(grid_rwb), dim3(kC1U8bThreads), 0, st, ra)
      if (C == 9) CONV1_U8_RWB_LAUNCH(9); else if (C == 6) CONV1_U8_RWB_LAUNCH(6); else CONV1_U8_RWB_LAUNCH(3);
#undef CONV1_U8_RWB_LAUNCH
      return curla_launch_status();
    }
#define CONV1_U8_RW_LAUNCH(CC) hipLaunchKernelGGL((conv1_u8_rw_fwd_kernel<CC>), dim3(grid_rw), dim3(kC1U8Threads), 0, st, ra)
    if (C == 9) CONV1_U8_RW_LAUNCH(9); else if (C == 12) CONV1_U8_RW_LAUNCH(12); else if (C == 6) CONV1_U8_RW_LAUNCH(6); else CONV1_U8_RW_LAUNCH(3);
#undef CONV1_U8_RW_LAUNCH
    return curla_launch_status();
  }
  if (src_kind == 1 && !(ABL_HOST & 128)) {
    // uint8 ring: the band stays bytes in LDS; the tallest band that leaves room for two workgroups per CU
    const int RSb = ((Wc * C + 15) & ~15) + 16;
    int th = a.Ho;
    while (th > 1 && (size_t)(2 * th + 1) * RSb > 76 * 1024) --th;
    a.nbands = (a.Ho + th - 1) / th;
    a.th = (a.Ho + a.nbands - 1) / a.nbands;  // near-equal bands
    size_t lds = (size_t)(2 * a.th + 1) * RSb + 32;
    const size_t wl8 = (size_t)3 * ((3 * C + 3) & ~3) * 32 * sizeof(float);  // the kernel's k-major weight image
    if (lds < wl8) lds = wl8;
    const int grid = (B + a.B2) * a.nbands;
#define CONV1_U8_LAUNCH(CC)                                                                                 \
  {                                                                                                         \
    rc = set_lds(conv1_fwd_u8_kernel<CC>, lds);                                                             \
    if (rc == CURLA_OK) hipLaunchKernelGGL((conv1_fwd_u8_kernel<CC>), dim3(grid), dim3(512), lds, st, a);   \
  }
    // one band = the whole crop in LDS: the hybrid form (row walk out of LDS) unless option conv1_u8 = band asks for the
    // banded loop (its staging gives a lane one 16-byte run of a crop row: rows of at most 64 runs)
    if (a.nbands == 1 && Wc * C <= 64 * 16 && curla_opt(kOptConv1U8) != 2) {
      rw::Geom G;
      G.Hi = Hc, G.Wi = Wc, G.Ho = a.Ho, G.Wo = a.Wo;
      rw::plan_units(G, a.Ho, a.Wo, 16);
#define CONV1_U8_WALK(CC)                                                                                     \
  {                                                                                                           \
    rc = set_lds(conv1_u8_walk_kernel<CC>, lds);                                                              \
    if (rc == CURLA_OK) hipLaunchKernelGGL((conv1_u8_walk_kernel<CC>), dim3(grid), dim3(512), lds, st, a, G); \
  }
      if (C == 9) CONV1_U8_WALK(9) else if (C == 12) CONV1_U8_WALK(12) else if (C == 6) CONV1_U8_WALK(6) else CONV1_U8_WALK(3)
#undef CONV1_U8_WALK
      if (rc != CURLA_OK) return rc;
      return curla_launch_status();
    }
    if (C == 9) CONV1_U8_LAUNCH(9) else if (C == 12) CONV1_U8_LAUNCH(12) else if (C == 6) CONV1_U8_LAUNCH(6) else CONV1_U8_LAUNCH(3)
#undef CONV1_U8_LAUNCH
    if (rc != CURLA_OK) return rc;
    return curla_launch_status();
  }
  if (src_kind == 2 && curla_opt(kOptConv1F32) == 0 && (long long)Hc * Wc * C * 4 < (1LL << 30)) {
    // float NHWC minibatch: row walk, nothing staged (conv1_rw.h)
    rw::Conv1Args ra;
    ra.src = static_cast<const float*>(src), ra.w = w, ra.bias = bias, ra.out = out;
    ra.B = B, ra.Hc = Hc, ra.Wc = Wc, ra.Ho = a.Ho, ra.Wo = a.Wo, ra.scale = scale;
    ra.g.Hi = Hc, ra.g.Wi = Wc, ra.g.Ho = a.Ho, ra.g.Wo = a.Wo;
    rw::plan_units(ra.g, a.Ho, a.Wo, 16);
    const int cap = curla_cu_count();
    const int grid_rw = B < cap ? B : cap;
#define CONV1_RW_LAUNCH(CC) hipLaunchKernelGGL((conv1_rw_fwd_kernel<CC>), dim3(grid_rw), dim3(512), 0, st, ra)
    if (C == 12) CONV1_RW_LAUNCH(12); else if (C == 9) CONV1_RW_LAUNCH(9); else if (C == 6) CONV1_RW_LAUNCH(6); else CONV1_RW_LAUNCH(3);
#undef CONV1_RW_LAUNCH
    return curla_launch_status();
  }
  // two workgroups per CU so one stages while the other computes
  a.th = plan_band_conv1(a.Ho, a.Wo, Wc, C, 0, 76 * 1024);
  a.nbands = (a.Ho + a.th - 1) / a.th;
  const int RS = ((Wc * C + 3) & ~3) + 4;
  size_t lds = ((size_t)(2 * a.th + 1) * RS + 8) * sizeof(float);
  if (lds < wl) lds = wl;
  const int nitems = B * a.nbands;
  const int grid = nitems < 2 * curla_cu_count() ? nitems : 2 * curla_cu_count();
  CURLA_DISPATCH_C(C, src_kind, CONV1_FWD_LAUNCH, grid, lds, st, a);
  if (rc != CURLA_OK) return rc;
  return curla_launch_status();
}

int curla_conv1_fwd(const void* src, int src_kind, const int64_t* idx, const int32_t* h1, const int32_t* w1,
                    const float* w, const float* bias, float* out, int B, int C, int Hs, int Ws, int Hc, int Wc,
                    int channels, float scale, void* stream) {
  return conv1_fwd_impl(src, src_kind, idx, h1, w1, w, bias, out, B, C, Hs, Ws, Hc, Wc, channels, scale, stream, nullptr);
}

int curla_conv1_fwd2(const uint8_t* ring, const int64_t* idx, const int32_t* h1, const int32_t* w1, const float* w,
                     const float* bias, float* out, int B, const int64_t* idx2, const int32_t* h1_2,
                     const int32_t* w1_2, const float* w2, const float* bias2, float* out2, int B2, int C, int Hs, int Ws,
                     int Hc, int Wc, int channels, float scale, void* stream) {
  CURLA_REQUIRE(w2 && bias2 && out2 && B2 > 0);
  Conv1Second sec{idx2, h1_2, w1_2, w2, bias2, out2, B2};
  return conv1_fwd_impl(ring, 1, idx, h1, w1, w, bias, out, B, C, Hs, Ws, Hc, Wc, channels, scale, stream, &sec);
}

// workspace (floats) the weight-gradient kernels need for their per-workgroup slabs
size_t curla_conv_wgrad_workspace_floats(int cin) {
  return (size_t)4 * curla_cu_count() * ((size_t)32 * cin * 9 + 32);  // at most four workgroups (slabs) per CU
}

// option s1_wgrad (options.h): Winograd F(3,2) along x, or in both directions (auto: a third fewer matrix instructions for
// ~40 instead of ~10 VALU instructions per step; alone 77 -> 69 us, beside the bf16x3 data gradient 125 -> 118 us for
// 512 samples of 35 x 35 gradients, 1179 -> 1080 us for 1024 of 79 x 79: tools/s1_bwd_bench.py)
static bool wgrad_two_d() { return curla_opt(kOptS1Wgrad) != 1; }

static rw::WgradArgs wgrad_args(const float* in, const float* g, float* workspace, int B, int Hi, int Wi, int Ho, int Wo) {
  const bool two_d = wgrad_two_d();
  return rw::WgradArgs{in, g, workspace, B, Hi, Wi, Ho, Wo, two_d ? rw::plan4p(Hi, Wi, Ho, Wo) : rw::plan4(Hi, Wi, Ho, Wo),
                       two_d ? 1 : 0};
}

static int launch_wgrad_s1(const float* in, const float* g, float* workspace, int B, int Hi, int Wi, int channels,
                           hipStream_t st, int* nslabs) {
  CURLA_REQUIRE(in && g && workspace && B > 0 && Hi >= 3 && Wi >= 3);
  if (channels != 32) return gen_wgrad_s1(in, g, workspace, B, Hi, Wi, channels, st, nslabs);
  if (!rw_supported(Hi, Wi)) return CURLA_ERR_UNSUPPORTED;
  CURLA_REQUIRE(aligned16(in) && aligned16(g));
  const rw::WgradArgs ra = wgrad_args(in, g, workspace, B, Hi, Wi, Hi - 2, Wi - 2);
  const int cap = 2 * curla_cu_count();
  const int grid = B < cap ? B : cap;
  const size_t lds = kPartialS1 * sizeof(float);
  int rc = set_lds(wgrad_rw_kernel, lds);
  if (rc != CURLA_OK) return rc;
  hipLaunchKernelGGL(wgrad_rw_kernel, dim3(grid), dim3(256), lds, st, ra);
  *nslabs = grid;
  return curla_launch_status();
}

int curla_conv3x3_s1_wgrad(const float* in, const float* g, float* dw, float* db, float* workspace, int B, int Hi,
                           int Wi, int channels, void* stream) {
  CURLA_REQUIRE(dw && db);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int grid = 0;
  int rc = launch_wgrad_s1(in, g, workspace, B, Hi, Wi, channels, st, &grid);
  if (rc != CURLA_OK) return rc;
  const int nw = channels * channels * 9;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((nw + channels + 31) / 32), dim3(1024), 0, st, workspace, grid, nw, channels,
                     dw, db);
  return curla_launch_status();
}

int curla_conv3x3_s1_wgrad_slabs(const float* in, const float* g, float* workspace, int B, int Hi, int Wi, int channels,
                                 int* nslabs, void* stream) {
  CURLA_REQUIRE(nslabs);
  return launch_wgrad_s1(in, g, workspace, B, Hi, Wi, channels, static_cast<hipStream_t>(stream), nslabs);
}

int curla_conv3x3_s1_bwd_slabs(const float* in, const float* g, const float* w, float* gin, float* workspace, int B, int Hi,
                               int Wi, int channels, int* nslabs, void* stream) {
  CURLA_REQUIRE(in && g && w && gin && workspace && nslabs && B > 0 && Hi >= 3 && Wi >= 3);
  if (channels != 32) {  // generic width: the two gradients as two launches
    hipStream_t gst = static_cast<hipStream_t>(stream);
    const int rc = gen_wgrad_s1(in, g, workspace, B, Hi, Wi, channels, gst, nslabs);
    return rc != CURLA_OK ? rc : gen_dgrad(g, w, in, gin, B, Hi - 2, Wi - 2, channels, gst);
  }
  CURLA_REQUIRE(aligned16(in) && aligned16(g) && aligned16(w) && aligned16(gin));
  const int Ho = Hi - 2, Wo = Wi - 2;
  if (!rw_supported(Hi, Wi)) return CURLA_ERR_UNSUPPORTED;
  if (dgrad_b3()) {
    // the data gradient on the bf16 matrix cores, beside the weight gradient's workgroups in one launch: one workgroup
    // of each kind per CU side by side (auto; with this data gradient it is the faster split for long launches too --
    // configs[4] 35.9 against 35.2 update()/s), or two and two (option bwd_split = 0)
    const int split_opt = curla_opt(kOptBwdSplit);
    const bool split2 = split_opt ? split_opt == 2 : true;
    const int cap2 = split2 ? curla_cu_count() : 2 * curla_cu_count();
    const int n2 = B < cap2 ? B : cap2;
    const rw::WgradArgs wr = wgrad_args(in, g, workspace, B, Hi, Wi, Ho, Wo);
    rw::Args dr = rw_dgrad_args(g, w, in, gin, B, Ho, Wo);
    dr.g[0] = rwb::plan(Ho, Wo, Ho + 2, Wo + 2);
    size_t lds2 = rwb::kWBytes;
    if (lds2 < kPartialS1 * sizeof(float)) lds2 = kPartialS1 * sizeof(float);
    int rc2 = set_lds(bwd_rwb2_kernel, lds2);
    if (rc2 != CURLA_OK) return rc2;
    hipLaunchKernelGGL(bwd_rwb2_kernel, dim3(2 * n2), dim3(256), lds2, static_cast<hipStream_t>(stream), wr, dr, n2);
    *nslabs = n2;
    return curla_launch_status();
  }
  if (dgrad_f43(Wi)) {
    // wide rows: the data gradient with Winograd F(4,3) (one wave per SIMD), the weight gradient as its own launch
    const int rcw = launch_wgrad_s1(in, g, workspace, B, Hi, Wi, channels, static_cast<hipStream_t>(stream), nslabs);
    if (rcw != CURLA_OK) return rcw;
    return launch_dgrad43(g, w, in, gin, B, Ho, Wo, static_cast<hipStream_t>(stream));
  }
  // Weight gradient and data gradient of the layer in ONE launch (both only read the layer's output gradient): the
  // first n workgroups run the weight-gradient body, the next n the data-gradient body, each owning samples k, k + n,
  // ... (the two do about the same number of MFMAs per sample).  Grid: 2 x CUs workgroups of each kind, run one kind
  // after the other -- or, for short launches, ONE workgroup of each kind per CU side by side: the layer takes as long
  // (a CU shared between the two kinds is no busier than one shared by two of a kind), but the weight gradient leaves
  // half as many slabs for the reduction to read.  Long launches keep the 2 + 2 form: there a few percent of imbalance
  // between the kinds (the tail runs at one workgroup per CU) costs more than the slabs (B = 1024 at 81 x 81: +0.3 ms
  // per update).  Option bwd_split (options.h) forces either form.
  const int split_opt = curla_opt(kOptBwdSplit);
  const bool split2 = split_opt ? split_opt == 2 : (long long)B * Ho * Wo <= (1LL << 20);
  const int cap2 = split2 ? curla_cu_count() : 2 * curla_cu_count();
  const int n2 = B < cap2 ? B : cap2;
  const rw::WgradArgs wr = wgrad_args(in, g, workspace, B, Hi, Wi, Ho, Wo);
  const rw::Args dr = rw_dgrad_args(g, w, in, gin, B, Ho, Wo);
  size_t lds2 = rw::kWFloats * sizeof(float);
  if (lds2 < kPartialS1 * sizeof(float)) lds2 = kPartialS1 * sizeof(float);
  int rc2 = set_lds(bwd_rw2_kernel, lds2);
  if (rc2 != CURLA_OK) return rc2;
  hipLaunchKernelGGL(bwd_rw2_kernel, dim3(2 * n2), dim3(256), lds2, static_cast<hipStream_t>(stream), wr, dr, n2);
  *nslabs = n2;
  return curla_launch_status();
}

int curla_wgrad_reduce_multi(int njobs, const float* const* slabs, const int* nslabs, const int* nw, const int* nb,
                             float* const* dw, float* const* db, void* stream) {
  CURLA_REQUIRE(njobs > 0 && njobs <= kMaxReduceJobs && slabs && nslabs && nw && dw && db);
  ReduceJobs J;
  int blocks = 0;
  for (int j = 0; j < njobs; ++j) {
    CURLA_REQUIRE(slabs[j] && dw[j] && db[j] && nslabs[j] > 0 && nw[j] > 0);
    J.partial[j] = slabs[j], J.dw[j] = dw[j], J.db[j] = db[j], J.nslabs[j] = nslabs[j], J.nw[j] = nw[j];
    J.nb[j] = nb ? nb[j] : 32;
    CURLA_REQUIRE(J.nb[j] > 0);
    J.first_block[j] = blocks;
    blocks += (nw[j] + J.nb[j] + 31) / 32;
  }
  J.first_block[njobs] = blocks;
  J.njobs = njobs;
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks), dim3(1024), 0, static_cast<hipStream_t>(stream), J);
  return curla_launch_status();
}

#define WGRAD1_LAUNCH(SRC, CC, grid, lds, st, a)                                        \
  {                                                                                     \
    rc = set_lds(wgrad1_kernel<SRC, CC>, lds);                                          \
    if (rc == CURLA_OK) hipLaunchKernelGGL((wgrad1_kernel<SRC, CC>), dim3(grid), dim3(512), lds, st, a); \
  }

static int launch_wgrad1(const void* src, int src_kind, const int64_t* idx, const int32_t* h1, const int32_t* w1,
                         const float* g, float* workspace, int B, int C, int Hs, int Ws, int Hc, int Wc, int channels,
                         float scale, void* stream, int* nslabs) {
  CURLA_REQUIRE(g && workspace);
  int rc = conv1_common_check(src, src_kind, B, C, Hs, Ws, Hc, Wc, h1, w1);
  if (rc != CURLA_OK) return rc;
  if (channels != 32)
    return gen_wgrad1(gen_src(src, src_kind, idx, h1, w1, C, Hs, Ws, Hc, Wc, scale), g, workspace, B, channels,
                      static_cast<hipStream_t>(stream), nslabs);
  CURLA_REQUIRE(aligned16(g));
  Wgrad1Args a;
  a.src = src, a.idx = idx, a.h1 = h1, a.w1 = w1, a.g = g, a.partial = workspace;
  a.B = B, a.C = C, a.Hs = Hs, a.Ws = Ws, a.Hc = Hc, a.Wc = Wc;
  a.Ho = (Hc - 3) / 2 + 1, a.Wo = (Wc - 3) / 2 + 1;
  a.scale = scale;
  a.lds_bytes = 0;
  a.dbg = ABL_HOST;
  const int nw = 32 * C * 9;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int grid;
  if (src_kind == 2 && curla_opt(kOptConv1F32) == 0 && (long long)Hc * Wc * C * 4 < (1LL << 30)) {
    // float NHWC minibatch: row walk, nothing staged (conv1_rw.h)
    rw::Wgrad1Args ra;
    ra.src = static_cast<const float*>(src), ra.g = g, ra.partial = workspace;
    ra.B = B, ra.Hc = Hc, ra.Wc = Wc, ra.Ho = a.Ho, ra.Wo = a.Wo, ra.scale = scale;
    ra.gg.Hi = Hc, ra.gg.Wi = Wc, ra.gg.Ho = a.Ho, ra.gg.Wo = a.Wo;
    rw::plan_units(ra.gg, a.Ho, a.Wo, 4);
    const int cap = curla_cu_count();
    grid = B < cap ? B : cap;
    const size_t lds = (size_t)(nw + 32) * sizeof(float);
#define WGRAD1_RW_LAUNCH(CC)                                                                     \
  {                                                                                              \
    rc = set_lds(wgrad1_rw_kernel<CC>, lds);                                                     \
    if (rc == CURLA_OK) hipLaunchKernelGGL((wgrad1_rw_kernel<CC>), dim3(grid), dim3(512), lds, st, ra); \
  }
    if (C == 12) WGRAD1_RW_LAUNCH(12) else if (C == 9) WGRAD1_RW_LAUNCH(9) else if (C == 6) WGRAD1_RW_LAUNCH(6) else WGRAD1_RW_LAUNCH(3)
#undef WGRAD1_RW_LAUNCH
    if (rc != CURLA_OK) return rc;
    *nslabs = grid;
    return curla_launch_status();
  }
  if (src_kind == 1 && !(ABL_HOST & 256)) {
    // uint8 ring, input band kept as bytes: two 512-thread workgroups per CU (<= 76 KB of LDS each).
    const int RSb = ((Wc * C + 15) & ~15) + 16;
    auto band_bytes = [&](int th) { return (size_t)(2 * th + 1) * RSb; };
    // (measured at 84x84x9, B = 512: 71.9 us with two 512-thread workgroups per CU, 77.4 us with four 256-thread ones:
    // the shorter bands' extra halo rows and slabs cost more than the finer interleaving buys)
    int nwaves = 8;
    size_t budget = 76 * 1024;
    if ((ABL_HOST & 1024) && band_bytes(4) <= 38 * 1024) nwaves = 4, budget = 38 * 1024;
    int th = a.Ho;
    while (th > 1 && band_bytes(th) > budget) --th;
    a.nbands = (a.Ho + th - 1) / th;
    a.th = (a.Ho + a.nbands - 1) / a.nbands;
    size_t lds = (((size_t)(2 * a.th + 1) * RSb + 15) & ~(size_t)15) + 32;
    if (lds < (size_t)(nw + 32) * sizeof(float)) lds = (size_t)(nw + 32) * sizeof(float);
    if (lds < (size_t)nwaves * 1024) lds = (size_t)nwaves * 1024;  // the final cross-wave sum: one tile of every wave
    {  // ... and all of a wave's tiles at once (ONE pass of the sum: 48.3 -> 44.8 us at 84x84x9) where two workgroups
       // of that size still share a CU
      const int k9 = 9 * C, ntiles = 2 * ((k9 % 16 == 1) ? k9 / 16 : (k9 + 15) / 16);
      const size_t one_pass = (size_t)ntiles * nwaves * 1024;
      if (nwaves == 8 && one_pass <= 80 * 1024 && lds < one_pass) lds = one_pass;
    }
    // round 6: the bf16-matrix-core form (wgrad1_u8b_kernel) wherever a lane's 8 consecutive pixels wrap at most once
    // (option wgrad1_u8: auto / b16; f32 keeps the f32-input MFMA)
    const bool b16 = curla_opt(kOptWgrad1U8) != 1 && nwaves == 8 && a.Wo >= 8 && 3 * C <= 32;  // (C = 12: nine tiles per
    // channel half do not fit four waves per SIMD)
    if (b16) {
      const int ntd = (3 * C + 15) / 16;
      const size_t one_pass = (size_t)(2 * 3 * ntd) * nwaves * 1024;  // all of a wave's tiles in ONE pass of the final sum
      if (one_pass <= 80 * 1024 && lds < one_pass) lds = one_pass;
    }
    a.lds_bytes = (unsigned)lds;
    const int nitems = B * a.nbands;
    const int per_cu = nwaves == 4 ? 4 : 2;
    grid = nitems < per_cu * curla_cu_count() ? nitems : per_cu * curla_cu_count();
#define WGRAD1_U8B_LAUNCH(CC)                                                                                      \
  {                                                                                                                \
    rc = set_lds(wgrad1_u8b_kernel<CC, 8>, lds);                                                                   \
    if (rc == CURLA_OK) hipLaunchKernelGGL((wgrad1_u8b_kernel<CC, 8>), dim3(grid), dim3(512), lds, st, a);         \
  }
    if (b16) {
      if (C == 9) WGRAD1_U8B_LAUNCH(9) else if (C == 6) WGRAD1_U8B_LAUNCH(6) else WGRAD1_U8B_LAUNCH(3)
      if (rc != CURLA_OK) return rc;
      *nslabs = grid;
      return curla_launch_status();
    }
#undef WGRAD1_U8B_LAUNCH
#define WGRAD1_U8_LAUNCH(CC)                                                                                       \
  {                                                                                                                \
    if (nwaves == 4) {                                                                                             \
      rc = set_lds(wgrad1_u8_kernel<CC, 4>, lds);                                                                  \
      if (rc == CURLA_OK) hipLaunchKernelGGL((wgrad1_u8_kernel<CC, 4>), dim3(grid), dim3(256), lds, st, a);        \
    } else {                                                                                                       \
      rc = set_lds(wgrad1_u8_kernel<CC, 8>, lds);                                                                  \
      if (rc == CURLA_OK) hipLaunchKernelGGL((wgrad1_u8_kernel<CC, 8>), dim3(grid), dim3(512), lds, st, a);        \
    }                                                                                                              \
  }
    if (C == 9) WGRAD1_U8_LAUNCH(9) else if (C == 12) WGRAD1_U8_LAUNCH(12) else if (C == 6) WGRAD1_U8_LAUNCH(6) else WGRAD1_U8_LAUNCH(3)
#undef WGRAD1_U8_LAUNCH
  } else {
    a.th = plan_band_conv1(a.Ho, a.Wo, Wc, C, 0, 150 * 1024);  // input rows only, one workgroup per CU
    a.nbands = (a.Ho + a.th - 1) / a.th;
    const int RS = ((Wc * C + 3) & ~3) + 4;
    size_t lds = ((size_t)(2 * a.th + 1) * RS + 8) * sizeof(float);
    if (lds < (size_t)(nw + 32) * sizeof(float)) lds = (size_t)(nw + 32) * sizeof(float);
    const int nitems = B * a.nbands;
    grid = nitems < curla_cu_count() ? nitems : curla_cu_count();
    CURLA_DISPATCH_C(C, src_kind, WGRAD1_LAUNCH, grid, lds, st, a);
  }
  if (rc != CURLA_OK) return rc;
  *nslabs = grid;
  return curla_launch_status();
}

int curla_conv1_wgrad(const void* src, int src_kind, const int64_t* idx, const int32_t* h1, const int32_t* w1,
                      const float* g, float* dw, float* db, float* workspace, int B, int C, int Hs, int Ws, int Hc,
                      int Wc, int channels, float scale, void* stream) {
  CURLA_REQUIRE(dw && db);
  int grid = 0;
  int rc = launch_wgrad1(src, src_kind, idx, h1, w1, g, workspace, B, C, Hs, Ws, Hc, Wc, channels, scale, stream, &grid);
  if (rc != CURLA_OK) return rc;
  const int nw = channels * C * 9;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((nw + channels + 31) / 32), dim3(1024), 0, static_cast<hipStream_t>(stream),
                     workspace, grid, nw, channels, dw, db);
  return curla_launch_status();
}

int curla_conv1_wgrad_slabs(const void* src, int src_kind, const int64_t* idx, const int32_t* h1, const int32_t* w1,
                            const float* g, float* workspace, int B, int C, int Hs, int Ws, int Hc, int Wc, int channels,
                            float scale, int* nslabs, void* stream) {
  CURLA_REQUIRE(nslabs);
  return launch_wgrad1(src, src_kind, idx, h1, w1, g, workspace, B, C, Hs, Ws, Hc, Wc, channels, scale, stream, nslabs);
}



}  // extern "C"
