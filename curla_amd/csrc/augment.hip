// Pixel augmentations that produce float observations (reference: augmentations.py:78-205).
//
// The reference implements ColorJiggle / NoisyCover with kornia (un-vendored, version un-pinned), so there is
// no reference arithmetic to pin against: the algorithm below is this build's own statement of kornia's
// documented behaviour (oracle/curla_oracle.py restates it on the CPU; parity is to that restatement only).
// Both kernels read the uint8 NHWC replay ring directly (gather by frame index fused in) and write the
// float32 NHWC minibatch [B][H][W][C] in [0,255] that the first conv kernel consumes (src kind 2).
#include "common.h"

namespace {

constexpr float kTwoPi = 6.283185307179586f;

// The colour-space round trips are the whole cost of the jitter kernel (VALU-bound: ~300 instructions per RGB pixel
// with IEEE divisions and integer modulos), so divisions are reciprocal multiplies (v_rcp_f32, 1 ulp) and the sector
// index uses the range the hue is known to be in; the differences to exact division are ~1e-7 relative.  Round 4: only
// the hue shift still makes the round trip; the saturation change is evaluated in RGB (jiggle_rgb).
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ void rgb_to_hsv(float r, float g, float b, float& h, float& s, float& v) {
  const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
  v = mx;
  float d = mx - mn;
  s = fminf(d * rcp_fast(mx + 1e-8f), 1.f);  // (the reciprocal's last-bit error must not push s past 1: v(1-s) >= 0)
  if (d == 0.f) d = 1.f;
  const float rc = mx - r, gc = mx - g, bc = mx - b;
  float hh;
  if (r == mx)          // first maximum wins, as argmax does
    hh = bc - gc;
  else if (g == mx)
    hh = (rc - bc) + 2.f * d;
  else
    hh = (gc - rc) + 4.f * d;
  hh = hh * rcp_fast(d) * (1.f / 6.f);
  hh = hh - floorf(hh);  // python-style % 1
  h = kTwoPi * hh;
}

// h in [0, 2 pi] (what rgb_to_hsv and the hue shift below produce)
__device__ __forceinline__ void hsv_to_rgb(float h, float s, float v, float& r, float& g, float& b) {
  const float h6 = h * (6.f / kTwoPi);
  // sector floor(h6) % 6; h6 == 6 (hue rounded up to 2 pi) is sector 5 with f = 1, the same colour as sector 0, f = 0
  const int hi = min(max((int)h6, 0), 5);
  const float f = h6 - (float)hi;
  const float p = v * (1.f - s), q = v * (1.f - f * s), t = v * (1.f - (1.f - f) * s);
  r = (hi == 0 || hi == 5) ? v : (hi == 1) ? q : (hi == 4) ? t : p;
  g = (hi == 1 || hi == 2) ? v : (hi == 0) ? t : (hi == 3) ? q : p;
  b = (hi == 3 || hi == 4) ? v : (hi == 2) ? t : (hi == 5) ? q : p;
}

// One RGB pixel in [0,1] through the jitter chain.  p = (apply, contrast, saturation, hue_radians) of the pixel's
// frame; o0..o3 = the call's permutation of {0 brightness (factor 0: identity), 1 contrast, 2 saturation, 3 hue}.
__device__ __forceinline__ void jiggle_rgb(float& r, float& g, float& bl, const float* p, int o0, int o1, int o2,
                                           int o3) {
  if (p[0] == 0.f) return;
  const float con = p[1], sat = p[2], hue = p[3];
#pragma unroll
  for (int step = 0; step < 4; ++step) {
    const int op = step == 0 ? o0 : step == 1 ? o1 : step == 2 ? o2 : o3;
    if (op == 1) {
      r = fminf(fmaxf(r * con, 0.f), 1.f);
      g = fminf(fmaxf(g * con, 0.f), 1.f);
      bl = fminf(fmaxf(bl * con, 0.f), 1.f);
    } else if (op == 2) {
      // Saturation WITHOUT the HSV round trip.  RGB -> HSV -> (s <- clamp(s sat)) -> RGB leaves v and h alone and maps
      // every channel c = v (1 - k s) (k in {0, f, 1 - f, 1} by the hue sector) to v (1 - k s'), i.e.
      // c' = v - (v - c) s' / s: three FMAs instead of ~50 instructions of sector arithmetic and selects -- the same
      // function up to rounding (the restatement in oracle/curla_oracle.py keeps the round trip; agreement 1e-6).
      const float mx = fmaxf(r, fmaxf(g, bl)), mn = fminf(r, fminf(g, bl));
      const float s = fminf((mx - mn) * rcp_fast(mx + 1e-8f), 1.f);
      const float s2 = fminf(fmaxf(s * sat, 0.f), 1.f);
      const float ratio = s > 0.f ? s2 * rcp_fast(s) : 0.f;
      r = fmaxf(mx - (mx - r) * ratio, 0.f);  // (s' / s can exceed (mx + 1e-8) / d by an ulp: keep [0, v])
      g = fmaxf(mx - (mx - g) * ratio, 0.f);
      bl = fmaxf(mx - (mx - bl) * ratio, 0.f);
    } else if (op == 3) {
      float h, s, v;
      rgb_to_hsv(r, g, bl, h, s, v);
      h = h + hue;
      h = h - kTwoPi * floorf(h * (1.f / kTwoPi));  // fmod into [0, 2pi)
      h = fminf(fmaxf(h, 0.f), kTwoPi);
      hsv_to_rgb(h, s, v, r, g, bl);
    }
  }
}

// params[img] = (apply, contrast, saturation, hue_radians); order[4] = permutation of {0 brightness(identity),
// 1 contrast, 2 saturation, 3 hue}; one image = one RGB frame of the stack (augmentations.py:124-128).
__global__ void color_jiggle_kernel(const uint8_t* frames, const int64_t* idx, const float* params, const int* order,
                                    int B, int C, int H, int W, float* out) {
  const int k = C / 3;
  const size_t n = (size_t)B * H * W * k;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const int o0 = order[0], o1 = order[1], o2 = order[2], o3 = order[3];
  for (; i < n; i += stride) {
    const int fr = i % k;
    const size_t pix = i / k;  // (b, y, x) flattened
    const int b = pix / ((size_t)H * W);
    const size_t yx = pix - (size_t)b * H * W;
    const int64_t fi = idx ? idx[b] : b;
    const uint8_t* src = frames + ((size_t)fi * H * W + yx) * C + 3 * fr;
    constexpr float k255 = 1.f / 255.f;
    float r = src[0] * k255, g = src[1] * k255, bl = src[2] * k255;  // `image_batch /= 255.0`, augmentations.py:118
    jiggle_rgb(r, g, bl, params + ((size_t)b * k + fr) * 4, o0, o1, o2, o3);
    float* dst = out + pix * C + 3 * fr;
    dst[0] = r * 255.f, dst[1] = g * 255.f, dst[2] = bl * 255.f;
  }
}

// The same with one thread per PIXEL (all K frames of the stack) and one grid row per sample: no 64-bit division per
// element (the flat form spends ~100 instructions on i % k, i / k, pix / (H W)), the pixel's 3 K bytes as aligned
// dwords when 3 K is a multiple of 4 (frame_stack 4: three dwords), its 3 K floats as 16-byte stores.  The arithmetic
// per RGB triple is the flat kernel's (jiggle_rgb), so the results are bit-identical.
template <int K>
__global__ __launch_bounds__(256) void color_jiggle_pixel_kernel(const uint8_t* frames, const int64_t* idx,
                                                                   const float* params, const int* order, int HW,
                                                                   float* out) {
  constexpr int C = 3 * K;
  const int b = blockIdx.y;
  const int p_raw = blockIdx.x * 256 + threadIdx.x;
  if ((p_raw & ~63) >= HW) return;             // (whole waves past the image leave; a partly covered wave stays whole:
  const int p = p_raw < HW ? p_raw : HW - 1;   //  its lanes past the image redo the last pixel and help with the stores)
  const int o0 = order[0], o1 = order[1], o2 = order[2], o3 = order[3];
  const int64_t fi = idx ? idx[b] : b;
  const uint8_t* src = frames + ((size_t)fi * HW + p) * C;
  uint8_t px[C];
  if (C % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0) {
#pragma unroll
    for (int u = 0; u < C / 4; ++u) {
      const uint32_t d = reinterpret_cast<const uint32_t*>(src)[u];
      px[4 * u] = d & 0xff, px[4 * u + 1] = (d >> 8) & 0xff, px[4 * u + 2] = (d >> 16) & 0xff, px[4 * u + 3] = d >> 24;
    }
  } else {
#pragma unroll
    for (int u = 0; u < C; ++u) px[u] = src[u];
  }
  float v[C];
  constexpr float k255 = 1.f / 255.f;
#pragma unroll
  for (int fr = 0; fr < K; ++fr) {
    float r = px[3 * fr] * k255, g = px[3 * fr + 1] * k255, bl = px[3 * fr + 2] * k255;
    jiggle_rgb(r, g, bl, params + ((size_t)b * K + fr) * 4, o0, o1, o2, o3);
    v[3 * fr] = r * 255.f, v[3 * fr + 1] = g * 255.f, v[3 * fr + 2] = bl * 255.f;
  }
  float* dst = out + ((size_t)b * HW + p) * C;
  if (C % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    // A lane holds its pixel's C floats: stored directly, a wave's store instruction would write 16 bytes per lane
    // C * 4 bytes apart (a third of every 128-byte line per instruction at C = 12).  Through LDS instead: the wave's
    // 64 pixels are 64 C / 4 consecutive 16-byte chunks of the output; lane l stores chunks l, l + 64, ... -- whole lines.
    __shared__ __attribute__((aligned(16))) float stage[4][64 * C];
    float* mine = stage[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int u = 0; u < C / 4; ++u)
      *reinterpret_cast<f32x4*>(mine + lane * C + 4 * u) = f32x4{v[4 * u], v[4 * u + 1], v[4 * u + 2], v[4 * u + 3]};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int p0 = blockIdx.x * 256 + (threadIdx.x & ~63);  // the wave's first pixel
    const int nvalid = min(64, HW - p0);                    // pixels of this wave inside the image (> 0: p < HW here)
    float* wave_out = out + ((size_t)b * HW + p0) * C;
#pragma unroll
    for (int u = 0; u < C / 4; ++u) {
      const int chunk = lane + 64 * u;  // 16-byte chunk of the wave's output: floats [4 chunk, 4 chunk + 4)
      if (4 * chunk < nvalid * C) *reinterpret_cast<f32x4*>(wave_out + 4 * chunk) = *reinterpret_cast<const f32x4*>(mine + 4 * chunk);
    }
  } else if (p_raw < HW) {
#pragma unroll
    for (int u = 0; u < C; ++u) dst[u] = v[u];
  }
}

// the same on the reference's tensor contract: float NCHW [B][C][H][W] in [0,255] in and out
// (ColorJiggle.training_augmentation(image_batch), augmentations.py:105-136; in == out is allowed)
__global__ void color_jiggle_nchw_kernel(const float* in, const float* params, const int* order, int B, int C, int H,
                                         int W, float* out) {
  const int k = C / 3;
  const size_t plane = (size_t)H * W;
  const size_t n = (size_t)B * k * plane;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const int o0 = order[0], o1 = order[1], o2 = order[2], o3 = order[3];
  for (; i < n; i += stride) {
    const size_t img = i / plane;  // b * k + frame
    const size_t yx = i - img * plane;
    const size_t base = img * 3 * plane + yx;
    constexpr float k255 = 1.f / 255.f;
    float r = in[base] * k255, g = in[base + plane] * k255, bl = in[base + 2 * plane] * k255;
    jiggle_rgb(r, g, bl, params + img * 4, o0, o1, o2, o3);
    out[base] = r * 255.f, out[base + plane] = g * 255.f, out[base + 2 * plane] = bl * 255.f;
  }
}

// rows [0,top) and [H-bottom,H) of every frame are painted with colors[c%3], then noise is added and the
// result clamped to [0,255] (augmentations.py:185-203)
__global__ void noisy_cover_kernel(const uint8_t* frames, const int64_t* idx, const float* noise, float c0, float c1,
                                   float c2, int top, int bottom, int B, int C, int H, int W, float* out) {
  const size_t n = (size_t)B * H * W * C;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int c = i % C;
    const size_t pix = i / C;
    const int b = pix / ((size_t)H * W);
    const size_t yx = pix - (size_t)b * H * W;
    const int y = yx / W;
    const int64_t fi = idx ? idx[b] : b;
    float v = (float)frames[((size_t)fi * H * W + yx) * C + c];
    if (y < top || y >= H - bottom) v = (c % 3 == 0) ? c0 : (c % 3 == 1) ? c1 : c2;
    v += noise[i];
    out[i] = fminf(fmaxf(v, 0.f), 255.f);
  }
}

// the same on the reference's tensor contract: float NCHW in [0,255] in, noise NCHW, out NCHW
// (NoisyCover.training_augmentation(image_batch), augmentations.py:170-205; in == out is allowed)
__global__ void noisy_cover_nchw_kernel(const float* in, const float* noise, float c0, float c1, float c2, int top,
                                        int bottom, int B, int C, int H, int W, float* out) {
  const size_t n = (size_t)B * C * H * W;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int y = (i / W) % H;
    const int c = (i / ((size_t)W * H)) % C;
    float v = in[i];
    if (y < top || y >= H - bottom) v = (c % 3 == 0) ? c0 : (c % 3 == 1) ? c1 : c2;
    v += noise[i];
    out[i] = fminf(fmaxf(v, 0.f), 255.f);
  }
}

// plain gather + u8 -> f32 (identity augmentation materialised as NHWC floats)
__global__ void gather_nhwc_kernel(const uint8_t* frames, const int64_t* idx, int B, size_t frame, float* out) {
  const size_t n = (size_t)B * frame;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int b = i / frame;
    const int64_t fi = idx ? idx[b] : b;
    out[i] = (float)frames[(size_t)fi * frame + (i - (size_t)b * frame)];
  }
}

inline int blocks_for(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b < 8192 ? b : 8192);
}

}  // namespace

extern "C" {

int curla_color_jiggle(const uint8_t* frames, const int64_t* idx, const float* params, const int32_t* order, int B,
                       int C, int H, int W, float* out, void* stream) {
  CURLA_REQUIRE(frames && params && order && out && B > 0 && C > 0 && C % 3 == 0 && H > 0 && W > 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int K = C / 3, HW = H * W;
  if (K >= 1 && K <= 4 && B <= 65535 && (long long)H * W < (1LL << 30)) {
    const dim3 grid((HW + 255) / 256, B);
    if (K == 4) hipLaunchKernelGGL(color_jiggle_pixel_kernel<4>, grid, dim3(256), 0, st, frames, idx, params, order, HW, out);
    else if (K == 3) hipLaunchKernelGGL(color_jiggle_pixel_kernel<3>, grid, dim3(256), 0, st, frames, idx, params, order, HW, out);
    else if (K == 2) hipLaunchKernelGGL(color_jiggle_pixel_kernel<2>, grid, dim3(256), 0, st, frames, idx, params, order, HW, out);
    else hipLaunchKernelGGL(color_jiggle_pixel_kernel<1>, grid, dim3(256), 0, st, frames, idx, params, order, HW, out);
    return curla_launch_status();
  }
  hipLaunchKernelGGL(color_jiggle_kernel, dim3(blocks_for((size_t)B * H * W * (C / 3))), dim3(256), 0, st, frames, idx,
                     params, order, B, C, H, W, out);
  return curla_launch_status();
}

int curla_noisy_cover(const uint8_t* frames, const int64_t* idx, const float* noise, float c0, float c1, float c2,
                      int top, int bottom, int B, int C, int H, int W, float* out, void* stream) {
  CURLA_REQUIRE(frames && noise && out && B > 0 && C > 0 && H > 0 && W > 0 && top >= 0 && bottom >= 0);
  hipLaunchKernelGGL(noisy_cover_kernel, dim3(blocks_for((size_t)B * H * W * C)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), frames, idx, noise, c0, c1, c2, top, bottom, B, C, H, W, out);
  return curla_launch_status();
}

int curla_color_jiggle_nchw(const float* in, const float* params, const int32_t* order, int B, int C, int H, int W,
                            float* out, void* stream) {
  CURLA_REQUIRE(in && params && order && out && B > 0 && C > 0 && C % 3 == 0 && H > 0 && W > 0);
  hipLaunchKernelGGL(color_jiggle_nchw_kernel, dim3(blocks_for((size_t)B * H * W * (C / 3))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, params, order, B, C, H, W, out);
  return curla_launch_status();
}

int curla_noisy_cover_nchw(const float* in, const float* noise, float c0, float c1, float c2, int top, int bottom,
                           int B, int C, int H, int W, float* out, void* stream) {
  CURLA_REQUIRE(in && noise && out && B > 0 && C > 0 && H > 0 && W > 0 && top >= 0 && bottom >= 0);
  hipLaunchKernelGGL(noisy_cover_nchw_kernel, dim3(blocks_for((size_t)B * H * W * C)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, noise, c0, c1, c2, top, bottom, B, C, H, W, out);
  return curla_launch_status();
}

int curla_gather_nhwc(const uint8_t* frames, const int64_t* idx, int B, int C, int H, int W, float* out, void* stream) {
  CURLA_REQUIRE(frames && out && B > 0 && C > 0 && H > 0 && W > 0);
  hipLaunchKernelGGL(gather_nhwc_kernel, dim3(blocks_for((size_t)B * H * W * C)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), frames, idx, B, (size_t)C * H * W, out);
  return curla_launch_status();
}

}  // extern "C"
