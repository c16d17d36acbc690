// Encoder convolutions for filter counts other than 32 (the reference is generic in num_filters: encoder.py:54-63,
// exposed as --num_filters at train.py:84; every BASELINE configuration uses 32, which the row-walk kernels of conv.hip
// are built for).  Plain direct convolutions on the vector ALU, one thread per output element, fp32 multiply-adds in a
// fixed order (bitwise reproducible): the CORRECTNESS path behind the same C entry points -- a CurlSacAgent with
// num_filters = 16 or 64 runs, is held to the oracle by tests/test_gpu_agent.py, and is one to two orders of magnitude
// slower per FLOP than the 32-filter path.  Included by conv.hip.
//
// Layouts as everywhere: activations NHWC fp32 [B][H][W][channels], weights OIHW in place in the parameter buffer; the
// first layer reads its input through the same three source kinds as curla_conv1_fwd (uint8 ring + index + crop
// offsets, float NCHW, float NHWC) and multiplies by `scale` (encoder.py:78: obs / 255).
#pragma once

namespace gen {

struct Src {
  const void* src;
  int kind;  // 1: uint8 ring [N][Hs][Ws][C]; 0: float NCHW [B][C][Hc][Wc]; 2: float NHWC [B][Hc][Wc][C]
  const int64_t* idx;
  const int32_t* h1;
  const int32_t* w1;
  int C, Hs, Ws, Hc, Wc;
  float scale;
};

// input pixel (y, x), channel c of sample b's (cropped) observation, scaled
__device__ __forceinline__ float first_in(const Src& s, int b, int y, int x, int c) {
  if (s.kind == 1) {
    const int64_t fi = s.idx ? s.idx[b] : (int64_t)b;
    const int oh = s.h1 ? s.h1[b] : 0, ow = s.w1 ? s.w1[b] : 0;
    const uint8_t* p = static_cast<const uint8_t*>(s.src);
    return (float)p[(((size_t)fi * s.Hs + oh + y) * s.Ws + ow + x) * s.C + c] * s.scale;
  }
  const float* p = static_cast<const float*>(s.src);
  if (s.kind == 0) return p[(((size_t)b * s.C + c) * s.Hc + y) * s.Wc + x] * s.scale;
  return p[(((size_t)b * s.Hc + y) * s.Wc + x) * s.C + c] * s.scale;
}

// out[b][y][x][co] = relu(bias[co] + sum_{dy, dx, ci} in(b, stride y + dy, stride x + dx, ci) w[co][ci][dy][dx])
// FIRST: the input is the observation (Src), stride 2; else `in` [B][Hi][Wi][Cin], stride 1.
template <bool FIRST>
__global__ __launch_bounds__(256) void conv_fwd_kernel(Src s, const float* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int B,
                                                       int Hi, int Wi, int Cin, int Ho, int Wo, int Cout) {
  const size_t total = (size_t)B * Ho * Wo * Cout;
  const int stride = FIRST ? 2 : 1;
  for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(o % Cout);
    size_t pix = o / Cout;
    const int x = (int)(pix % Wo);
    pix /= Wo;
    const int y = (int)(pix % Ho), b = (int)(pix / Ho);
    float acc = bias[co];
    for (int ci = 0; ci < Cin; ++ci) {
      const float* wk = w + ((size_t)co * Cin + ci) * 9;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int yy = stride * y + dy, xx = stride * x + dx;
          const float v = FIRST ? first_in(s, b, yy, xx, ci) : in[(((size_t)b * Hi + yy) * Wi + xx) * Cin + ci];
          acc = fmaf(v, wk[dy * 3 + dx], acc);
        }
    }
    out[o] = acc > 0.f ? acc : 0.f;
  }
}

// Data gradient of a stride-1 layer: gin[b][y][x][ci] = [act_below > 0] sum_{dy, dx, co} g[b][y - dy][x - dx][co]
// w[co][ci][dy][dx] over the positions inside g's Ho x Wo (gin is (Ho + 2) x (Wo + 2)).
__global__ __launch_bounds__(256) void conv_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                         const float* __restrict__ act_below, float* __restrict__ gin,
                                                         int B, int Ho, int Wo, int C) {
  const int Hi = Ho + 2, Wi = Wo + 2;
  const size_t total = (size_t)B * Hi * Wi * C;
  for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(o % C);
    size_t pix = o / C;
    const int x = (int)(pix % Wi);
    pix /= Wi;
    const int y = (int)(pix % Hi), b = (int)(pix / Hi);
    float acc = 0.f;
    if (act_below[o] > 0.f) {
      for (int dy = 0; dy < 3; ++dy) {
        const int yy = y - dy;
        if (yy < 0 || yy >= Ho) continue;
        for (int dx = 0; dx < 3; ++dx) {
          const int xx = x - dx;
          if (xx < 0 || xx >= Wo) continue;
          const float* gp = g + (((size_t)b * Ho + yy) * Wo + xx) * C;
          for (int co = 0; co < C; ++co) acc = fmaf(gp[co], w[((size_t)co * C + ci) * 9 + dy * 3 + dx], acc);
        }
      }
    }
    gin[o] = acc;
  }
}

// Weight gradient: ONE slab [Cout * Cin * 9 | Cout] (the layout curla_wgrad_reduce_multi sums; here there is one).
// Workgroup (co, ci) -- and, behind those, one per co for the bias -- walks the B Ho Wo output positions in strides of
// its 256 threads, nine accumulators per thread, then adds the threads' sums in a fixed tree.
template <bool FIRST>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(Src s, const float* __restrict__ in, const float* __restrict__ g,
                                                         float* __restrict__ slab, int B, int Hi, int Wi, int Cin, int Ho,
                                                         int Wo, int Cout) {
  __shared__ float sm[9][256];
  const int tid = threadIdx.x;
  const int stride = FIRST ? 2 : 1;
  const int npos = B * Ho * Wo;
  const int nwb = Cout * Cin;
  float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool is_bias = (int)blockIdx.x >= nwb;
  const int co = is_bias ? (int)blockIdx.x - nwb : (int)blockIdx.x / Cin;
  const int ci = is_bias ? 0 : (int)blockIdx.x % Cin;
  for (int n = tid; n < npos; n += 256) {
    const float gv = g[(size_t)n * Cout + co];
    if (is_bias) {
      acc[0] += gv;
      continue;
    }
    if (gv == 0.f) continue;  // (ReLU-masked gradients: exact zeros contribute exact zeros)
    const int x = n % Wo, y = (n / Wo) % Ho, b = n / (Wo * Ho);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int yy = stride * y + dy, xx = stride * x + dx;
        const float v = FIRST ? first_in(s, b, yy, xx, ci) : in[(((size_t)b * Hi + yy) * Wi + xx) * Cin + ci];
        acc[dy * 3 + dx] = fmaf(gv, v, acc[dy * 3 + dx]);
      }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) sm[k][tid] = acc[k];
  __syncthreads();
  for (int half = 128; half >= 1; half >>= 1) {
    if (tid < half)
#pragma unroll
      for (int k = 0; k < 9; ++k) sm[k][tid] += sm[k][tid + half];
    __syncthreads();
  }
  if (is_bias) {
    if (tid == 0) slab[(size_t)nwb * 9 + co] = sm[0][0];
  } else if (tid < 9) {
    slab[((size_t)co * Cin + ci) * 9 + tid] = sm[tid][0];
  }
}

inline int grid_for(size_t total) {
  const size_t b = (total + 255) / 256;
  return (int)(b < 65535 * 16 ? b : 65535 * 16);
}

inline bool channels_ok(int channels) { return channels >= 4 && channels <= 256 && channels % 4 == 0; }

}  // namespace gen
